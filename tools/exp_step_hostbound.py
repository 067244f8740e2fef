"""Is the host-stepped rate bound by the host (Python + ctypes + launch) or by the GPU?  Times env.step() at a tiny
and at the full env count, eagerly and as a captured hipGraph of 100 steps (no host work per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hex_gym_env_amd as hx

N = int(os.environ.get("N", 11))
for E in (64, 4096, 16384, 65536):
    env = hx.HexVecEnv(N, E, encoding="B", seed=1)
    for _ in range(5000):
        env.step()
    torch.cuda.synchronize()

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n, 1e6 * t_issue / n

    eager = [timed(env.step, 2000) for _ in range(3)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(100):
                env.step()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    gr = [timed(g.replay, 20) for _ in range(3)]
    print("E=%6d  eager: %s us/launch by events (host issue %s us/call);  graph of 100 steps: %s us/launch" % (
        E, " ".join("%.2f" % a for a, _ in eager), " ".join("%.2f" % b for _, b in eager),
        " ".join("%.2f" % (a / 100) for a, _ in gr)), flush=True)
