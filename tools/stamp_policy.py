"""Diagnostic: where a wavefront of k_mlp64_mfma (32 boards) spends its cycles.
    python tools/stamp_policy.py build      (in the build container: csrc/variants/pol_stamps.so with -DHV_POL_STAMPS)
    HEXVEC_MLP_NOSPLIT=1 HEXVEC_SO=hex_gym_env_amd/csrc/variants/pol_stamps.so python tools/stamp_policy.py   (on the GPU
    box; the stamps are in k_mlp64_mfma, the one-wavefront-per-tile kernel)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hex_gym_env_amd", "csrc")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.join(CSRC, "variants"), exist_ok=True)
    objs = []
    for f in ("hexvec", "hexvec_sample", "hexvec_policy", "hexvec_wpb", "hexvec_qpb_s", "hexvec_qpb_m", "hexvec_qpb_l", "hexvec_qpb_x", "hexvec_lpb64"):
        o = os.path.join(CSRC, f + ".o")
        if f == "hexvec_policy":
            o = os.path.join(CSRC, "variants", "policy_stamps.o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                   "-DHV_POL_STAMPS", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", "-o", o,
                                   os.path.join(CSRC, f + ".hip")])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                           os.path.join(CSRC, "variants", "pol_stamps.so")] + objs)
    sys.exit(0)

sys.path.insert(0, ROOT)
import numpy as np
import torch
from hex_gym_env_amd import _lib
from hex_gym_env_amd.vec_selfplay import MlpActorPolicy, mlp_actions

SPLIT = os.environ.get("HEXVEC_MLP_NOSPLIT") is None   # stamps of k_mlp64_mfma2 (two wavefronts per tile): rows 2 * tile + w
names = ["rows requested + written to LDS", "layer 1 MFMAs", "tanh 1", "layer 2 MFMAs", "tanh 2", "layer 3 MFMAs + epilogues", "combine + store"]
for n, E in ((7, 32), (7, 32768), (7, 262144), (11, 32), (11, 65536), (11, 262144), (13, 16384)):
    Cc = n * n
    pol = MlpActorPolicy(Cc).cuda()
    obs = (torch.randint(0, 3, (E, Cc), device="cuda") - 1).to(torch.int8)
    legal = (obs == 0).to(torch.uint8)
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(5):
        mlp_actions(pol.packed(), Cc, 64, 64, obs, legal, rng=(st, 1, 0, False))
    torch.cuda.synchronize()
    nt = min(E // 32, 8192)
    if SPLIT and 2 * (E // 32) * 2 <= 3 * 1024:
        nt = min(2 * (E // 32), 8192)
    buf = np.zeros((nt, 10), np.uint64)
    assert _lib.lib().hexvec_debug_policy_stamps(buf.ctypes.data_as(C.c_void_p), nt) == 0
    d = np.diff(buf[:, :8].astype(np.int64), axis=1).astype(np.float64)
    span = float(buf[:, 7].max() - buf[:, 0].min())
    print("%dx%d, %d boards: s_memtime ticks per wavefront (mean over %d); first start to last end of these: %.0f"
          % (n, n, E, nt, span))
    for nm, v in zip(names, d.mean(0)):
        print("   %-34s %9.0f" % (nm, v))
    print("   %-34s %9.0f" % ("total", d.sum(1).mean()))
