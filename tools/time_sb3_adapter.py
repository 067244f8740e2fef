#!/usr/bin/env python3
"""HexSB3VecEnv.step (numpy in, numpy out: what SB3's collect_rollouts calls) per second, run on the GPU box"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hex_gym_env_amd.sb3_adapter import HexSB3VecEnv

for n, E in ((5, 64), (11, 64), (11, 1024), (11, 8192), (7, 32768)):
    env = HexSB3VecEnv(board_size=n, num_envs=E, seed=1)
    env.reset()
    rng = np.random.default_rng(0)
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        m = env.action_masks()
        a = (rng.random(m.shape) * m).argmax(1)
        env.step(a)
        k += 1
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(50):
        env.step(a * 0 + (env.action_masks().argmax(1)))
    per = (time.perf_counter() - t1) / 50
    print("%2dx%-2d %6d envs: %8.0f VecEnv steps/s = %6.2f M env-steps/s with a numpy random policy; step() alone %.0f us"
          % (n, n, E, k / dt, k * E / dt / 1e6, per * 1e6), flush=True)
