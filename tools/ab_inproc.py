"""A/B of library builds INSIDE one process on the SAME output buffers (the speed level of a store-bound rollout
follows where its output buffers lie, so builds timed in different processes are not comparable):
    python tools/ab_inproc.py a.so b.so [...]     env: N (11), E (65536), T (100), SETS (3 buffer sets), OUT (1)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hex_gym_env_amd as hx
from hex_gym_env_amd import _lib

N, E, T = int(os.environ.get("N", 11)), int(os.environ.get("E", 65536)), int(os.environ.get("T", 100))
C = N * N
TWO = os.environ.get("TWO", "0") == "1"      # the two-sided rollout (W2) instead of the self-play one
envs = []
for so in sys.argv[1:]:
    _lib._lib, _lib.SO_PATH = None, os.path.abspath(so)        # the next HexVecEnv binds this build
    envs.append((os.path.basename(so), hx.HexVecEnv(N, E, encoding="B", seed=1, two_sided=TWO,
                                                    rollout_layout=os.environ.get("LAYOUT", "rollout") == "rollout")))
nsets = int(os.environ.get("SETS", 3))
sets = []
for i in range(nsets):
    sets.append({k: torch.empty((T, E) + ((C,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
                 for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32),
                               ("reward", torch.float32), ("done", torch.uint8))} if os.environ.get("OUT", "1") == "1" else {})
reps = max(10, 4000 // T)
for rnd in range(2):
    for i, out in enumerate(sets):
        line = "round %d buffers %d:" % (rnd, i)
        for name, env in envs:
            fn = env.rollout_random if TWO else env.rollout_selfplay
            for _ in range(reps):
                fn(T, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn(T, out)
            e1.record()
            torch.cuda.synchronize()
            line += "  %s %.3f" % (name, 1e3 * e0.elapsed_time(e1) / reps / T)
        print(line + "  us/step", flush=True)
