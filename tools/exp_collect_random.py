"""exp_collect_random.py -- the collection step with the reference's INITIAL pool (BaseRandomPolicy members, SelfplayWrapper.py:58-60)
and a random agent: masked sampling through hexvec_sample_masked instead of the actor kernel.  7x7, 32 768 boards."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, RandomLegalPolicy

dev = torch.device("cuda", 0)
N, E = int(os.environ.get("N", 7)), int(os.environ.get("E", 32768))
env = VecSelfPlayEnv(board_size=N, num_envs=E, buffer_size=20, seed=1, device=dev)
env.reset()
agent = RandomLegalPolicy(env._gen)
run = env.capture_collect_step(agent)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    for _ in range(50):
        run()
    torch.cuda.synchronize()
reps, inner = 20, 50
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for r in range(reps):
    for _ in range(inner):
        run()
    ev[r + 1].record()
torch.cuda.synchronize()
us = sorted(1e3 * ev[r].elapsed_time(ev[r + 1]) / inner for r in range(reps))
print(f"random pool, random agent, {N}x{N} x {E}: {us[reps // 2]:.2f} us per collection step (min {us[0]:.2f} max {us[-1]:.2f})")
