"""same-box A/B of library builds on the host-stepped step (one hexvec_step launch per env step):
    python tools/ab_step.py a.so b.so [...]     env: N (11), E (65536)"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import torch
    import hex_gym_env_amd as hx
    N, E = int(os.environ.get("N", 11)), int(os.environ.get("E", 65536))
    env = hx.HexVecEnv(N, E, encoding="B", seed=1)
    for _ in range(20000):
        env.step()
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(1000):
            env.step()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1))
    print("%-32s %s us/launch  (%s)" % (os.path.basename(os.environ["HEXVEC_SO"]), " ".join("%.2f" % r for r in res), env.last_kernel()), flush=True)
    sys.exit(0)

for rnd in range(2):
    for so in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, HEXVEC_SO=os.path.abspath(so)))
