"""experiment: rollout kernels with / without outputs (compute vs output cost)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx

N, E = 11, int(os.environ.get("E", 65536))
C = N * N
dev = torch.device("cuda", 0)


def timeit(fn, n=40):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for two in (True, False):
    T = int(os.environ.get("T", 64 if two else 50))
    env = hx.HexVecEnv(N, E, encoding="B", two_sided=two, seed=1)
    full = dict(obs=torch.empty((T, E, C), dtype=torch.int8, device=dev),
                legal=torch.empty((T, E, C), dtype=torch.uint8, device=dev),
                actions=torch.empty((T, E), dtype=torch.int32, device=dev),
                reward=torch.empty((T, E), dtype=torch.float32, device=dev),
                done=torch.empty((T, E), dtype=torch.uint8, device=dev))
    fn = env.rollout_random if two else env.rollout_selfplay
    for rnd in range(int(os.environ.get("ROUNDS", 3))):
      for name, out in (("all", full), ("none", {})):
        ms = timeit(lambda: fn(T, out))
        print(os.path.basename(os.environ.get("HEXVEC_SO", "libhexvec.so")), "%s %-9s %.3f ms/launch  %.2f G units/s" % ("W2" if two else "W1", name, ms, E * T / ms / 1e6))
    if not two:
        ms = timeit(lambda: env.step(), n=300)
        print(os.path.basename(os.environ.get("HEXVEC_SO", "libhexvec.so")), "host-stepped step: %.2f us/step  %.2f G steps/s" % (ms * 1e3, E / ms / 1e6))
    env.close()
