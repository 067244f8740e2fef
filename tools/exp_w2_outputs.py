"""experiment: two-sided rollout (W2) with all / no outputs -> how much is compute, how much the store path"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx
N, E, T = int(os.environ.get("N", 11)), 65536, 64
C = N * N
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True, two_sided=True)
full = {k: torch.empty((T, E) + ((C,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
        for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32), ("reward", torch.float32), ("done", torch.uint8))}
def timeit(out, n=40):
    for _ in range(20): env.rollout_random(T, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): env.rollout_random(T, out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / T * 1e3
for rnd in range(2):
    print("all outputs      %.3f us/ply" % timeit(full))
    print("obs + legal only %.3f us/ply" % timeit({k: full[k] for k in ("obs", "legal")}))
    print("scalars only     %.3f us/ply" % timeit({k: full[k] for k in ("actions", "reward", "done")}))
    print("none             %.3f us/ply" % timeit({}))
