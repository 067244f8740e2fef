"""experiment: rollout speed over the first seconds of sustained load (clock ramp / power management)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx
N, E, T = int(os.environ.get("N", 11)), 65536, int(os.environ.get("T", 20))
C = N * N
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True)
out = {k: torch.empty((T, E) + ((C,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
       for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32), ("reward", torch.float32), ("done", torch.uint8))}
env.rollout_selfplay(T, out); torch.cuda.synchronize()
t_start = time.perf_counter()
rows = []
while time.perf_counter() - t_start < float(os.environ.get("SECONDS", 4)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 200
    for _ in range(n): env.rollout_selfplay(T, out)
    e1.record(); torch.cuda.synchronize()
    rows.append((time.perf_counter() - t_start, e0.elapsed_time(e1) / n / T * 1e3))
for i in range(0, len(rows), max(1, len(rows) // 24)):
    print("t = %5.2f s: %.3f us/step" % rows[i])
print("last: %.3f us/step; min %.3f max %.3f" % (rows[-1][1], min(r[1] for r in rows), max(r[1] for r in rows)))
