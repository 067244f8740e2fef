"""Diagnostic (stamps build, tools/stamp_rollout.py build): where the dispatcher puts the playing wavefronts of the
self-play rollout -- XCC / SE / CU / SIMD of every workgroup's wave 0 -- and whether a workgroup's duration follows
its placement (two playing wavefronts on one SIMD) or its boards.
    HEXVEC_SO=hex_gym_env_amd/csrc/variants/stamps.so python tools/exp_placement.py"""
import ctypes as C
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import hex_gym_env_amd as hx

N, E, T = int(os.environ.get("N", 11)), int(os.environ.get("E", 65536)), int(os.environ.get("T", 20))
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True)
out = {k: torch.empty((T, E) + ((N * N,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
       for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32),
                     ("reward", torch.float32), ("done", torch.uint8))}
for _ in range(300):
    env.rollout_selfplay(T, out)
torch.cuda.synchronize()
L = hx._lib.lib()
nb = E // 64
durs, places = [], []
for rep in range(6):
    env.rollout_selfplay(T, out)
    torch.cuda.synchronize()
    st = np.zeros((nb, 12), np.uint64)
    ph = np.zeros((nb, 4), np.uint64)
    assert L.hexvec_debug_stamps(st.ctypes.data_as(C.c_void_p), nb) == 0
    assert L.hexvec_debug_phases(ph.ctypes.data_as(C.c_void_p), nb) == 0
    hw, xcc = st[:, 7].astype(np.int64), st[:, 3].astype(np.int64) & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    phf = ph.astype(np.float64)
    dur = phf[:, 3] - phf[:, 0]
    steps = phf[:, 2] - phf[:, 1]
    durs.append(steps)
    places.append((cuid, simd))
    per_cu = Counter(cuid.tolist())
    per_simd = Counter(zip(cuid.tolist(), simd.tolist()))
    share = np.array([per_simd[(c, s)] for c, s in zip(cuid.tolist(), simd.tolist())])
    print("rep %d: %d CUs used, workgroups per CU %s; playing wavefronts per (CU, SIMD): %s" % (
        rep, len(per_cu), dict(Counter(per_cu.values())), dict(Counter(per_simd.values()))))
    for k in sorted(set(share.tolist())):
        m = share == k
        print("   playing wavefront shares its SIMD with %d other(s): %4d workgroups, steps phase mean %.0f  p50 %.0f  max %.0f ticks (10 ns)"
              % (k - 1, m.sum(), steps[m].mean(), np.median(steps[m]), steps[m].max()))
    wpc = np.array([per_cu[c] for c in cuid.tolist()])
    for k in sorted(set(wpc.tolist())):
        m = wpc == k
        print("   on a CU with %d workgroups: %4d workgroups, steps phase mean %.0f max %.0f" % (k, m.sum(), steps[m].mean(), steps[m].max()))
    print("   steps phase by XCC (mean / max): " + "  ".join("%d: %.0f / %.0f" % (x, steps[xcc == x].mean(), steps[xcc == x].max())
                                                              for x in sorted(set(xcc.tolist()))))
    cyc = st[:, [0, 1, 2, 4, 5, 6]].astype(np.float64).sum(1)          # s_memtime ticks of the step loop (shader clock)
    print("   shader clock by XCC (loop cycles / loop time, GHz): " + "  ".join(
        "%d: %.2f" % (x, (cyc[xcc == x] / (steps[xcc == x] * 10.0)).mean()) for x in sorted(set(xcc.tolist())))
        + "   cycles per step by XCC: " + "  ".join("%d: %.0f" % (x, cyc[xcc == x].mean() / T) for x in sorted(set(xcc.tolist()))))
    print("   launch: first in to last out %.0f; slowest 5 workgroups: %s" % (
        phf[:, 3].max() - phf[:, 0].min(), [(int(i), int(steps[i]), int(share[i]), int(wpc[i])) for i in np.argsort(-steps)[:5]]))
d = np.stack(durs)
print("correlation of a workgroup's steps-phase duration between consecutive launches (same boards, placement may differ): %.2f"
      % np.corrcoef(d[:-1].ravel(), d[1:].ravel())[0, 1])
same = [(places[i][0] == places[i + 1][0]).mean() for i in range(len(places) - 1)]
print("fraction of workgroups on the same CU in consecutive launches: %s" % np.round(same, 2))
