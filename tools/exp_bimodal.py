"""experiment: is the 13x13 rollout's two-speed behaviour tied to buffer addresses or to the process?"""
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
big_o = torch.empty(T * E * C + (1 << 22), dtype=torch.int8, device="cuda")
big_l = torch.empty(T * E * C + (1 << 22), dtype=torch.uint8, device="cuda")
rest = {k: torch.empty((T, E), dtype=dt, device="cuda") for k, dt in (("actions", torch.int32), ("reward", torch.float32), ("done", torch.uint8))}
for _ in range(30): env.rollout_selfplay(T, dict(obs=big_o[:T*E*C].view(T,E,C), legal=big_l[:T*E*C].view(T,E,C), **rest))
for off in (0, 16, 256, 1024, 4096, 65536, 1 << 20, 0):
    out = dict(obs=big_o[off:off + T * E * C].view(T, E, C), legal=big_l[off:off + T * E * C].view(T, E, C), **rest)
    print("offset %8d: %.3f us/step (obs ptr %% 4096 = %d)" % (off, timeit(lambda: env.rollout_selfplay(T, out)), out["obs"].data_ptr() % 4096), flush=True)
for rep in range(4):
    env = hx.HexVecEnv(N, E, encoding="B", seed=1 + rep, rollout_layout=True)
    out = dict(obs=big_o[:T*E*C].view(T,E,C), legal=big_l[:T*E*C].view(T,E,C), **rest)
    print("fresh env %d: %.3f us/step" % (rep, timeit(lambda: env.rollout_selfplay(T, out))), flush=True)
