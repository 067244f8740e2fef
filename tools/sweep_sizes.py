"""rollout throughput per board size (E = 65 536, enc B): W1 self-play steps/s, W2 plies/s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx

E = 65536
dev = torch.device("cuda", 0)
for n in (5, 7, 9, 11, 13, 14):
    C = n * n
    row = [n]
    for two, T in ((False, 100), (True, 64)):
        env = hx.HexVecEnv(n, E, encoding="B", two_sided=two, seed=1)
        out = dict(obs=torch.empty((T, E, C), dtype=torch.int8, device=dev),
                   legal=torch.empty((T, E, C), dtype=torch.uint8, device=dev),
                   actions=torch.empty((T, E), dtype=torch.int32, device=dev),
                   reward=torch.empty((T, E), dtype=torch.float32, device=dev),
                   done=torch.empty((T, E), dtype=torch.uint8, device=dev))
        fn = env.rollout_random if two else env.rollout_selfplay
        for _ in range(5):
            fn(T, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(T, out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        rate = E * T / ms / 1e6
        row += [rate, rate * 1e9 * (2 * C + 9) / 1e12]
        env.close()
        del out
    print("N=%2d  W1 %.2f G steps/s (%.2f TB/s)   W2 %.2f G plies/s (%.2f TB/s)" % tuple(row))
