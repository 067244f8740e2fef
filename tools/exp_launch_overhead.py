"""experiment: per-launch cost of the self-play rollout vs steps per launch, with and without the
observation / mask outputs (fixed cost = state in/out + kernel boundary + tail; slope = one env step)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx

N, E = int(os.environ.get("N", 11)), int(os.environ.get("E", 65536))
C = N * N
dev = torch.device("cuda", 0)


def timeit(fn, n=200):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for lay in (True, False):
    env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=lay)
    Tm = 100
    full = dict(obs=torch.empty((Tm, E, C), dtype=torch.int8, device=dev),
                legal=torch.empty((Tm, E, C), dtype=torch.uint8, device=dev),
                actions=torch.empty((Tm, E), dtype=torch.int32, device=dev),
                reward=torch.empty((Tm, E), dtype=torch.float32, device=dev),
                done=torch.empty((Tm, E), dtype=torch.uint8, device=dev))
    for _ in range(30):
        env.rollout_selfplay(Tm, full)
    rows = []
    for T in (1, 2, 5, 10, 20, 50, 100):
        a = timeit(lambda: env.rollout_selfplay(T, full), n=max(20, 400 // T))
        b = timeit(lambda: env.rollout_selfplay(T, {}), n=max(20, 400 // T))
        rows.append((T, a, b))
        print("%s T=%3d: %8.1f us/launch with outputs (%.2f us/step), %8.1f us without" % (
            "rollout-layout" if lay else "bitboard", T, a, a / T, b), flush=True)
    (t1, a1, _), (t2, a2, _) = rows[3], rows[-1]
    slope = (a2 - a1) / (t2 - t1)
    print("  slope %.3f us/step, intercept %.1f us" % (slope, a1 - slope * t1))
    env.close()
