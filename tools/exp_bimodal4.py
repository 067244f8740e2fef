"""13x13 / 65 536 rollout: does the speed level follow the PHYSICAL placement of the output buffers?  One process,
several output-buffer sets allocated one after the other (all kept alive, so each lands on different physical pages),
each timed in turn, twice round; plus the same handle state throughout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx

N, E, T = int(os.environ.get("N", 13)), 65536, 100
C = N * N
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True)
sets, pads = [], []
for i in range(6):
    sets.append({k: torch.empty((T, E) + ((C,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
                 for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32),
                               ("reward", torch.float32), ("done", torch.uint8))})
    pads.append(torch.empty((i + 1) * 37 * 2 ** 20 + 4096 * i, dtype=torch.uint8, device="cuda"))   # shift the next set
for rnd in range(2):
    for i, out in enumerate(sets):
        for _ in range(40):
            env.rollout_selfplay(T, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            env.rollout_selfplay(T, out)
        e1.record()
        torch.cuda.synchronize()
        print("round %d, buffer set %d (obs at 0x%x): %.3f us/step" % (rnd, i, out["obs"].data_ptr(), 1e3 * e0.elapsed_time(e1) / 40 / T), flush=True)
