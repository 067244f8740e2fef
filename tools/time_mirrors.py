#!/usr/bin/env python3
"""Time the single-board drop-in mirrors (hex_gym_env_amd.hex_game / hex_single_game /
selfplay_wrapper) on the GPU box with the workloads of tools/time_reference.py (which times the
Python reference itself in the build container): random policy on both sides, the legal mask
fetched every agent step as ActionMasker would, one env, `--seconds` per point.

    python tools/time_mirrors.py [--board 5 11] [--seconds 3]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def engine_a(n, seconds):
    from hex_gym_env_amd.hex_game import HexGame, player, random_policy
    plies, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        sim = HexGame(player.BLACK, player.EMPTY * np.ones((n, n)), player.BLACK)
        while not sim.done:
            sim.make_move(random_policy(sim.board))
            plies += 1
    return plies / (time.perf_counter() - t0)


def env_a(n, seconds):
    from hex_gym_env_amd.hex_game import HexEnv, player, random_policy
    env = HexEnv(opponent_policy=random_policy, player_color=player.BLACK, board_size=n)
    steps, t0 = 0, time.perf_counter()
    env.reset()
    while time.perf_counter() - t0 < seconds:
        mask = env.get_action_mask()
        a = random.choice(np.flatnonzero(mask))
        _, _, done, _, _ = env.step(a)
        steps += 1
        if done:
            env.reset()
    return steps / (time.perf_counter() - t0)


def selfplay(n, seconds):
    from hex_gym_env_amd.hex_single_game import HexEnv
    from hex_gym_env_amd.selfplay_wrapper import selfplay_wrapper
    env = selfplay_wrapper(HexEnv)(board_size=n, buffer_size=4, scores=np.zeros(4))
    env.buffer.verbose = False
    steps, t0 = 0, time.perf_counter()
    env.reset()
    while time.perf_counter() - t0 < seconds:
        mask = env.legal_actions()
        a = random.choice(np.flatnonzero(mask))
        _, _, done, _, _ = env.step(a)
        steps += 1
        if done:
            env.reset()
    return steps / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, nargs="+", default=[5, 11])
    ap.add_argument("--seconds", type=float, default=3.0)
    args = ap.parse_args()
    random.seed(17)
    for n in args.board:
        for name, fn in (("engine A make_move + random_policy (plies/s)", engine_a),
                         ("hex-v0 HexEnv.step + get_action_mask (env-steps/s)", env_a),
                         ("SelfPlayEnv.step + legal_actions (env-steps/s)", selfplay)):
            fn(n, 0.3)
            print(json.dumps({"board": n, "workload": name, "processes": 1,
                              "mirror_per_s": round(fn(n, args.seconds))}), flush=True)


if __name__ == "__main__":
    main()
