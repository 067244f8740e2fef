"""experiment: does the 13x13 rollout's speed depend on the RELATIVE placement of the obs and legal buffers?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx
N, E, T = int(os.environ.get("N", 13)), 65536, 50
C = N * N
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / T * 1e3
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True)
sz = T * E * C
big = torch.empty(2 * sz + (1 << 26), dtype=torch.int8, device="cuda")      # ONE allocation holds both arrays
rest = {k: torch.empty((T, E), dtype=dt, device="cuda") for k, dt in (("actions", torch.int32), ("reward", torch.float32), ("done", torch.uint8))}
print("base ptr %% 2MiB = %d" % (big.data_ptr() % (1 << 21)))
for gap in (0, 256, 1024, 4096, 8192, 65536, 1 << 20, 1 << 21, (1 << 21) + 4096, 1 << 24, 3 << 23):
    out = dict(obs=big[:sz].view(T, E, C), legal=big[sz + gap:2 * sz + gap].view(torch.uint8).view(T, E, C), **rest)
    d = (out["legal"].data_ptr() - out["obs"].data_ptr())
    print("gap %9d: %.3f us/step   (legal - obs) %% 4096 = %4d, %% 2MiB = %7d" % (gap, timeit(lambda: env.rollout_selfplay(T, out)), d % 4096, d % (1 << 21)), flush=True)
for only in ("obs", "legal"):
    out = dict(**{only: big[:sz].view(torch.int8 if only == "obs" else torch.uint8).view(T, E, C)}, **rest)
    print("only %s: %.3f us/step" % (only, timeit(lambda: env.rollout_selfplay(T, out))))
