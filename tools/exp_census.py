"""diagnostic (stamps build): when and where the workgroups of one 13x13 rollout launch start"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hex_gym_env_amd as hx
N, E, T = int(os.environ.get("N", 13)), 65536, 50
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True)
out = {k: torch.empty((T, E) + ((N * N,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
       for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32), ("reward", torch.float32), ("done", torch.uint8))}
for _ in range(10): env.rollout_selfplay(T, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env.rollout_selfplay(T, out); e1.record(); torch.cuda.synchronize()
L = hx._lib.lib(); nb = E // 64
buf = np.zeros((nb, 12), np.uint64)
assert L.hexvec_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
start = buf[:, 11].astype(np.int64); start -= start.min()
hw = buf[:, 7].astype(np.int64)
cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print("launch %.1f us (%.3f us/step); workgroup start times (us after the first): p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
    e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / T, *(np.percentile(start, q) / 100.0 for q in (50, 90, 99, 100))))
cyc = buf[:, :7].astype(np.float64).sum(1)
print("shader clock while the launch ran: %.0f MHz (mean wavefront cycles / launch time)" % (cyc.mean() / (e0.elapsed_time(e1) * 1e3)))
late = start > 1000      # started more than 10 us after the first: a second pass
print("workgroups starting > 10 us late: %d of %d" % (late.sum(), nb))
key = se * 32 + sh * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct (SE,SH,CU) ids seen: %d; workgroups per id: min %d max %d; histogram %s" % (len(u), cnt.min(), cnt.max(), dict(zip(*np.unique(cnt, return_counts=True)))))
