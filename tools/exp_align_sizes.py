"""Is the store-bound rollout slower where a workgroup's 64-board output chunk is not a whole number of 128-byte
lines (64 x N*N bytes: odd N = half a line over)?  us per step and achieved TB/s by board size, several buffer sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx

E, T = 65536, 100
for N in (10, 11, 12, 13, 14):
    C = N * N
    env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True)
    res = []
    keep = []
    for i in range(4):
        out = {k: torch.empty((T, E) + ((C,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
               for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32),
                             ("reward", torch.float32), ("done", torch.uint8))}
        keep.append(out)
        for _ in range(30):
            env.rollout_selfplay(T, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            env.rollout_selfplay(T, out)
        e1.record()
        torch.cuda.synchronize()
        res.append(1e3 * e0.elapsed_time(e1) / 30 / T)
    for _ in range(30):
        env.rollout_selfplay(T, {})
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        env.rollout_selfplay(T, {})
    e1.record()
    torch.cuda.synchronize()
    noout = 1e3 * e0.elapsed_time(e1) / 30 / T
    print("%2dx%-2d chunk %5d B (%s a line multiple): %s us/step = %s TB/s; without outputs %.2f us/step" % (
        N, N, 64 * C, "is" if (64 * C) % 128 == 0 else "NOT", " ".join("%.2f" % r for r in res),
        " ".join("%.2f" % (E * (2 * C + 9) / r / 1e6) for r in res), noout), flush=True)
    del env, keep
    torch.cuda.empty_cache()
