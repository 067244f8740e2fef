"""Diagnostic: timeline of one host-stepped launch of the bitboard kernel (k_qpb<N,1,step>), per wavefront.
    python tools/build_variant.py qstamps unit=hexvec_qpb_m -DHV_QSTAMPS       (build container)
    HEXVEC_SO=hex_gym_env_amd/csrc/variants/qstamps.so python tools/stamp_step.py  (GPU box)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import hex_gym_env_amd as hx

N, E = int(os.environ.get("N", 11)), int(os.environ.get("E", 65536))
env = hx.HexVecEnv(N, E, encoding="B", seed=1)
for _ in range(3000):
    env.step()
torch.cuda.synchronize()
L = hx._lib.lib()
bpw = 64 // {1: 1, 2: 2, 3: 4, 4: 4}.get((N * N + 31) // 32, 8 if N * N <= 256 else 16)
nb = min(8192, (E + bpw - 1) // bpw)
q = np.percentile
for rep in range(3):
    for _ in range(20):
        env.step()
    torch.cuda.synchronize()
    ph = np.zeros((nb, 4), np.uint64)
    assert L.hexvec_debug_qphases(ph.ctypes.data_as(C.c_void_p), nb) == 0
    ph = ph.astype(np.float64)
    ph = ph[(ph > 0).all(1)]
    t0 = ph[:, 0].min()
    print("launch (%s): %d wavefronts stamped; first in to last out %.2f us" % (env.last_kernel(), len(ph), (ph[:, 3].max() - t0) / 100))
    for name, a in (("entry", ph[:, 0] - t0), ("state loaded", ph[:, 1] - t0), ("step computed", ph[:, 2] - t0), ("stores retired", ph[:, 3] - t0)):
        print("   %-15s (us after the first entry) p1 %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (
            name, *(q(a, x) / 100 for x in (1, 10, 50, 90, 99)), a.max() / 100))
    for name, a in (("load", ph[:, 1] - ph[:, 0]), ("compute", ph[:, 2] - ph[:, 1]), ("store", ph[:, 3] - ph[:, 2])):
        print("   per wavefront %-8s mean %.2f p10 %.2f p50 %.2f p90 %.2f us" % (name, a.mean() / 100, *(q(a, x) / 100 for x in (10, 50, 90))))
