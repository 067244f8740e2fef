"""host side of a host-stepped step at a small env count (the kernel is ~4 us): wall time per HexVecEnv.step() / split step"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx
for n, E in ((7, 64), (7, 8192), (11, 1024)):
    env = hx.HexVecEnv(n, E, encoding="B", seed=1)
    env.reset()
    for _ in range(200):
        env.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000):
        env.step()
    torch.cuda.synchronize()
    a = (time.perf_counter() - t0) / 5000 * 1e6
    act = torch.zeros(E, dtype=torch.int32, device="cuda")
    t0 = time.perf_counter()
    for _ in range(3000):
        env.step_agent(act); env.step_opponent(act)
    torch.cuda.synchronize()
    b = (time.perf_counter() - t0) / 3000 * 1e6
    print("%dx%d %5d envs: step() %.2f us per call; step_agent + step_opponent %.2f us per pair" % (n, n, E, a, b))
