"""Diagnostic: where a self-play rollout step of the rollout-layout kernel spends its cycles.
Builds libhexvec with -DHV_STAMPS into csrc/variants/stamps.so (s_memtime stamps around the sections of
a step; the product build has none), runs one rollout launch and prints cycles per section and step.
    python tools/stamp_rollout.py build      (in the build container)
    HEXVEC_SO=hex_gym_env_amd/csrc/variants/stamps.so python tools/stamp_rollout.py   (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hex_gym_env_amd", "csrc")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.join(CSRC, "variants"), exist_ok=True)
    objs = []
    for f in ("hexvec", "hexvec_sample", "hexvec_policy", "hexvec_wpb", "hexvec_qpb_s", "hexvec_qpb_m", "hexvec_qpb_l", "hexvec_qpb_x",
              "hexvec_lpb64"):
        o = os.path.join(CSRC, f + ".o")
        if f == "hexvec_lpb64":
            o = os.path.join(CSRC, "variants", "lpb64_stamps.o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                   "-DHV_STAMPS", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", "-o", o,
                                   os.path.join(CSRC, f + ".hip")])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                           os.path.join(CSRC, "variants", "stamps.so")] + objs)
    sys.exit(0)

sys.path.insert(0, ROOT)
import numpy as np
import torch
import hex_gym_env_amd as hx

N, E, T = int(os.environ.get("N", 11)), 65536, 100
env = hx.HexVecEnv(N, E, encoding="B", seed=1, rollout_layout=True)
out = {k: torch.empty((T, E) + ((N * N,) if k in ("obs", "legal") else ()), dtype=dt, device="cuda")
       for k, dt in (("obs", torch.int8), ("legal", torch.uint8), ("actions", torch.int32),
                     ("reward", torch.float32), ("done", torch.uint8))}
for _ in range(20):
    env.rollout_selfplay(T, out)
torch.cuda.synchronize()
L = hx._lib.lib()
nb = E // 64
buf = np.zeros((nb, 12), np.uint64)
assert L.hexvec_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
per = buf.astype(np.float64).mean(0) / T
names = ["agent action (rng + k-th empty)", "wait: helper snapshot (barrier 2)", "agent ply (place)",
         "finished boards after the agent", "opponent action + ply", "finished boards 2 / slot 3 / scalar stores",
         "publish emission (barrier 1)", "loop head",
         "  inside place (both plies): reads issued + bit / image updates", "  inside place: chase to the roots",
         "  inside place: union (keys, max, 8 byte writes)", "  (unused)"]
print("s_memtime ticks (shader cycles; the stamps themselves add ~15 %%) per step and playing wavefront, mean over %d wavefronts" % nb)
for n, v in zip(names, per):
    print("  %-62s %8.1f" % (n, v))
print("  %-45s %8.1f" % ("total", per[:8].sum()))

# launch phases (s_memtime at kernel entry / state in LDS / before the store-back / stores retired), per workgroup
T2 = int(os.environ.get("T", 20))
out2 = {k: torch.empty((T2,) + tuple(v.shape[1:]), dtype=v.dtype, device="cuda") for k, v in out.items()}
for _ in range(5):
    env.rollout_selfplay(T2, out2)
torch.cuda.synchronize()
ph = np.zeros((nb, 4), np.uint64)
assert L.hexvec_debug_phases(ph.ctypes.data_as(C.c_void_p), nb) == 0
ph = ph.astype(np.float64)
ph = ph[(ph > 0).all(1)]
print("(%d of %d workgroups carry all four stamps)" % (len(ph), nb))
span = ph[:, 3].max() - ph[:, 0].min()
print("launch of %d steps: first workgroup in to last out %.0f ticks of 10 ns; per workgroup (mean): state load %.0f, steps %.0f, "
      "store-back %.0f; entry spread %.0f, exit spread %.0f"
      % (T2, span, (ph[:, 1] - ph[:, 0]).mean(), (ph[:, 2] - ph[:, 1]).mean(), (ph[:, 3] - ph[:, 2]).mean(),
         ph[:, 0].max() - ph[:, 0].min(), ph[:, 3].max() - ph[:, 3].min()))
q = np.percentile
print("   entry times relative to the first: p50 %.0f p90 %.0f p99 %.0f; exit times relative to the last: p50 %.0f p10 %.0f p1 %.0f"
      % (q(ph[:, 0], 50) - ph[:, 0].min(), q(ph[:, 0], 90) - ph[:, 0].min(), q(ph[:, 0], 99) - ph[:, 0].min(),
         ph[:, 3].max() - q(ph[:, 3], 50), ph[:, 3].max() - q(ph[:, 3], 10), ph[:, 3].max() - q(ph[:, 3], 1)))
