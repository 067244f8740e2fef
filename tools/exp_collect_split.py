"""exp_collect_split.py -- config 3's collection (7x7, 32 768 boards, MLP agent + 20-slot MLP pool): one VecSelfPlayEnv of E
boards against K independent ones of E / K boards whose captured graphs replay concurrently on K streams.
The four launches of a step depend on each other and each is latency-bound at this batch; independent parts overlap their
latencies and boundaries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy

dev = torch.device("cuda", 0)
N, E, NS = int(os.environ.get("N", 7)), int(os.environ.get("E", 32768)), int(os.environ.get("NS", 8))
torch.manual_seed(0)
agent, opp = MlpActorPolicy(N * N).to(dev), MlpActorPolicy(N * N).to(dev)


def timed(fn, inner=8, reps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        fn()
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        for _ in range(inner):
            fn()
        ev[r + 1].record()
    torch.cuda.synchronize()
    us = sorted(1e3 * ev[r].elapsed_time(ev[r + 1]) / inner for r in range(reps))
    return us[reps // 2], us[0], us[-1]


for K in (1, 2, 4, 2, 1):
    envs = [VecSelfPlayEnv(board_size=N, num_envs=E // K, buffer_size=20, base_model=opp, seed=1 + k, device=dev) for k in range(K)]
    for e in envs:
        e.reset()
    runs = [e.capture_collect(agent, NS) for e in envs]
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]

    def both():
        cur = torch.cuda.current_stream(dev)
        if K == 1:
            runs[0]()
            return
        for s, r in zip(streams, runs):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                r()
        for s in streams:
            cur.wait_stream(s)
    med, lo, hi = timed(both)
    print(f"K={K}: {med / NS:.2f} us per step of all {E} boards (min {lo / NS:.2f} max {hi / NS:.2f}); degenerate {sum(e.degenerate_rows(agent) for e in envs)}", flush=True)
    del runs, envs
