"""Is the captured collection step (bench.py's collect_step leg) bound by the GPU or by the host side of run()?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy

dev = torch.device("cuda", 0)
n3, E3 = 7, 32768
torch.manual_seed(0)
agent, opp = MlpActorPolicy(n3 * n3).to(dev), MlpActorPolicy(n3 * n3).to(dev)
env3 = VecSelfPlayEnv(board_size=n3, num_envs=E3, buffer_size=20, base_model=opp, seed=1, device=dev)
env3.reset()
run = env3.capture_collect_step(agent)
for _ in range(3000):
    run()
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(2000):
        run()
    e1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("run(): %.1f us per step by events, host issue %.1f us per call" % (1e3 * e0.elapsed_time(e1) / 2000, 1e6 * t_issue / 2000))
g = env3._graph
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(2000):
        g.replay()
    e1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("bare graph.replay(): %.1f us per step by events, host issue %.1f us per call" % (1e3 * e0.elapsed_time(e1) / 2000, 1e6 * t_issue / 2000))
