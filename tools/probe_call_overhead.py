"""host-side cost of one HexVecEnv.step() call at small env counts"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx

for E in (64, 1024, 16384):
    env = hx.HexVecEnv(11, E, encoding="B", seed=1)
    a = torch.zeros(E, dtype=torch.int32, device="cuda")
    for _ in range(200):
        env.step()
    torch.cuda.synchronize()
    for name, fn in (("step()", lambda: env.step()), ("step(actions)", lambda: env.step(a))):
        t0 = time.perf_counter()
        n = 3000
        for _ in range(n):
            fn()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("E=%6d %-14s host issue %.2f us/call, wall %.2f us/call" % (E, name, t_issue / n * 1e6, t_all / n * 1e6))
