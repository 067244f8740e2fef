"""13x13 / 65 536 self-play rollout timed continuously in ONE process: does the speed of this store-bound kernel step
between two levels with TIME (memory / fabric clock state of the box) rather than with the process?  Prints us/step
every ~1.5 s next to the clocks rocm-smi reports (read-only query in a child process)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx

N, E, T = int(os.environ.get("N", 13)), 65536, 200
C = N * N
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True)
out = {k: torch.empty((T, E) + ((C,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
       for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32),
                     ("reward", torch.float32), ("done", torch.uint8))}


def clocks():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.split(":", 1)[1].strip() for l in r.splitlines() if "GPU[0]" in l and any(k in l for k in ("mclk", "fclk", "sclk", "socclk"))]
        return " | ".join(keep)
    except Exception as e:
        return "rocm-smi: %r" % (e,)


print("before:", clocks(), flush=True)
t_start = time.time()
k = 0
while time.time() - t_start < float(os.environ.get("SECONDS", 45)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(12):
        env.rollout_selfplay(T, out)
    e1.record()
    torch.cuda.synchronize()
    k += 1
    if k % 150 == 0 or k < 4:
        print("t = %5.1f s: %.3f us/step   %s" % (time.time() - t_start, 1e3 * e0.elapsed_time(e1) / 12 / T, clocks() if k % 300 == 0 else ""), flush=True)
print("after:", clocks())
