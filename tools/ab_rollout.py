"""same-box A/B of library builds on the rollout kernels:
    python tools/ab_rollout.py a.so b.so [...]      (run on the GPU box; builds alternate, 3 rounds)
env: N (11), E (65536), T (100), LAYOUT (rollout|bitboard), TWO (0|1)"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import torch
    import hex_gym_env_amd as hx
    N, E, T = int(os.environ.get("N", 11)), int(os.environ.get("E", 65536)), int(os.environ.get("T", 100))
    two = os.environ.get("TWO", "0") == "1"
    env = hx.HexVecEnv(N, E, encoding="B", seed=1, two_sided=two, rollout_layout=os.environ.get("LAYOUT", "rollout") == "rollout")
    C = N * N
    out = {k: torch.empty((T, E) + ((C,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
           for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32),
                         ("reward", torch.float32), ("done", torch.uint8))}
    if os.environ.get("OUT", "1") == "0":       # no outputs: the playing wavefront's own time per step
        out = {}
    fn = env.rollout_random if two else env.rollout_selfplay
    for _ in range(60):
        fn(T, out)
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn(T, out)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 40 / T * 1e3)
    print("%-40s %s us/step  (%s)" % (os.path.basename(os.environ["HEXVEC_SO"]), " ".join("%.3f" % r for r in res),
                                      env.last_kernel()), flush=True)
    sys.exit(0)

for rnd in range(int(os.environ.get("ROUNDS", 3))):
    for so in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, HEXVEC_SO=os.path.abspath(so)))
