"""13x13 / 65 536 rollout: obs and legal rows carved from ONE allocation at a chosen distance: is the speed level a
function of the distance between the two output streams (memory-channel aliasing), or of where the slab itself lies?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx

N, E, T = int(os.environ.get("N", 13)), 65536, 100
C = N * N
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True)
S = (T * E * C + 2 ** 21 - 1) // 2 ** 21 * 2 ** 21
small = {k: torch.empty((T, E), dtype=dt, device="cuda") for k, dt in (("actions", torch.int32), ("reward", torch.float32), ("done", torch.uint8))}


def timed(out):
    for _ in range(30):
        env.rollout_selfplay(T, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        env.rollout_selfplay(T, out)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 30 / T


slabs = []
for s in range(3):
    slab = torch.empty(2 * S + 2 ** 28, dtype=torch.uint8, device="cuda")
    slabs.append(slab)
    base = slab.data_ptr()
    for d in (0, 256, 4096, 65536, 2 ** 20, 2 ** 21 + 4096, 37 * 2 ** 20, 2 ** 27 + 2 ** 12):
        obs = slab[:T * E * C].view(torch.int8).view(T, E, C)
        legal = slab[S + d:S + d + T * E * C].view(T, E, C)
        out = dict(small, obs=obs, legal=legal)
        print("slab %d (0x%x): legal - obs = %d MiB + %7d B: %.3f us/step" % (s, base, S >> 20, d, timed(out)), flush=True)
    # one stream only
    print("slab %d: obs only %.3f, legal only %.3f us/step" % (s, timed(dict(small, obs=obs)), timed(dict(small, legal=legal))), flush=True)
