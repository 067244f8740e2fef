"""experiment: rollout speed vs (legal - obs) address difference modulo powers of two"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx
N, E, T = int(os.environ.get("N", 13)), 65536, int(os.environ.get("T", 50))
two = os.environ.get("TWO", "0") == "1"
C = N * N
def timeit(fn, n=12):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / T * 1e3
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True, two_sided=two)
fn = env.rollout_random if two else env.rollout_selfplay
sz = T * E * C
szr = (sz + (1 << 21) - 1) >> 21 << 21
big = torch.empty(2 * szr + (1 << 25), dtype=torch.int8, device="cuda")
base = (-big.data_ptr()) % (1 << 21)          # start obs on a 2 MiB boundary
rest = {k: torch.empty((T, E), dtype=dt, device="cuda") for k, dt in (("actions", torch.int32), ("reward", torch.float32), ("done", torch.uint8))}
for g in [0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 3 << 19, 5 << 10, 3 << 10, 12345 * 16]:
    o = big[base:base + sz].view(T, E, C)
    l = big[base + szr + g:base + szr + g + sz].view(torch.uint8).view(T, E, C)
    out = dict(obs=o, legal=l, **rest)
    print("delta %% 2MiB = %8d: %.3f us/step" % ((l.data_ptr() - o.data_ptr()) % (1 << 21), timeit(lambda: fn(T, out))), flush=True)
