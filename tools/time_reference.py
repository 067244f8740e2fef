#!/usr/bin/env python3
"""Time the Python reference itself in the BUILD container.

The reference cannot travel to the GPU box, so its own speed is measured here, on this
container's CPUs, with the method of BASELINE.md: reference files imported unmodified from
/root/reference (gymnasium / pygame stand-ins, as in tests/golden/make_fixtures.py), random
policy on both sides, the legal mask computed every agent step as ActionMasker would, one env
per process, `--procs` processes, `--seconds` per point.  Skips when /root/reference is absent.

    python tools/time_reference.py [--board 11] [--seconds 4] [--procs 1 8]
"""
import argparse
import json
import multiprocessing as mp
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _setup():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_fixtures
    make_fixtures._install_stubs()
    sys.path.insert(0, REF)


def _engine_a(args):
    n, seconds, seed = args
    _setup()
    from minihex import random_policy
    from minihex.HexGame import HexGame, player
    random.seed(seed)
    plies, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        sim = HexGame(player.BLACK, player.EMPTY * np.ones((n, n)), player.BLACK)
        while not sim.done:
            sim.make_move(random_policy(sim.board))
            plies += 1
    return plies / (time.perf_counter() - t0)


def _env_a(args):
    n, seconds, seed = args
    _setup()
    from minihex import random_policy
    from minihex.HexGame import HexEnv, player
    random.seed(seed)
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


def _selfplay(args):
    n, seconds, seed = args
    _setup()
    from minihex.HexSingleGame import HexEnv
    from minihex.SelfplayWrapper import selfplay_wrapper
    random.seed(seed)
    env = selfplay_wrapper(HexEnv)(board_size=n, buffer_size=4, scores=np.zeros(4))
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
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--procs", type=int, nargs="+", default=[1, os.cpu_count() or 1])
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print("reference not present: nothing to time")
        return
    out = []
    for n in args.board:
        for name, fn in (("engine A make_move + random_policy (plies/s)", _engine_a),
                         ("hex-v0 HexEnv.step + get_action_mask (env-steps/s)", _env_a),
                         ("SelfPlayEnv.step + legal_actions (env-steps/s)", _selfplay)):
            for p in args.procs:
                with mp.Pool(p) as pool:
                    rates = pool.map(fn, [(n, args.seconds, 17 + i) for i in range(p)])
                row = {"board": n, "workload": name, "processes": p, "total_per_s": round(sum(rates))}
                out.append(row)
                print(json.dumps(row), flush=True)
    print(json.dumps({"host": os.uname().nodename, "cpus": os.cpu_count(), "rows": len(out)}))


if __name__ == "__main__":
    main()
