"""experiment: where k_mlp64_mfma's time goes -- sampled vs arg-max (no noise), few vs many boards"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hex_gym_env_amd.vec_selfplay import MlpActorPolicy, mlp_actions

dev = torch.device("cuda", 0)


def ev(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n, Es in ((7, (32, 4096, 32768, 65536, 131072, 262144)), (11, (32, 65536, 131072, 262144)), (13, (16384, 65536))):
    C = n * n
    pol = MlpActorPolicy(C).to(dev)
    w = pol.packed()
    for E in Es:
        obs = (torch.randint(0, 3, (E, C), device=dev) - 1).to(torch.int8)
        legal = (obs == 0).to(torch.uint8)
        st = torch.zeros(2, dtype=torch.int64, device=dev)
        out = torch.empty(E, dtype=torch.int32, device=dev)
        ts = ev(lambda: mlp_actions(w, C, 64, 64, obs, legal, rng=(st, 1, 0, False), out=out))
        td = ev(lambda: mlp_actions(w, C, 64, 64, obs, legal, deterministic=True, rng=(st, 1, 0, False), out=out))
        flops = 2.0 * E * (C * 64 + 64 * 64 + 64 * C)
        print("%2dx%-2d %7d boards: sampled %6.1f us (%5.1f TFLOP/s)  arg-max %6.1f us" % (n, n, E, ts, flops / ts / 1e6, td), flush=True)
