"""Build an experiment variant of libhexvec into csrc/variants/<name>.so (same-box A/B with tools/ab_rollout.py):
    python tools/build_variant.py <name> [unit=hexvec_lpb64] [-DFLAG ...]
Only the named translation unit is recompiled with the extra flags; the other objects are the product's."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hex_gym_env_amd", "csrc")
name = sys.argv[1]
units = [a.split("=", 1)[1] for a in sys.argv[2:] if a.startswith("unit=")] or ["hexvec_lpb64"]
flags = [a for a in sys.argv[2:] if a.startswith("-")]
os.makedirs(os.path.join(CSRC, "variants"), exist_ok=True)
subprocess.check_call(["make", "-j8", "-C", CSRC, "libhexvec.so"], stdout=subprocess.DEVNULL)
objs = []
for f in ("hexvec", "hexvec_sample", "hexvec_policy", "hexvec_wpb", "hexvec_qpb_s", "hexvec_qpb_m", "hexvec_qpb_l",
          "hexvec_qpb_x", "hexvec_lpb64"):
    o = os.path.join(CSRC, f + ".o")
    if f in units:
        o = os.path.join(CSRC, "variants", f"{f}_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + flags +
                              ["-c", "-o", o, os.path.join(CSRC, f + ".hip")])
    objs.append(o)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                       os.path.join(CSRC, "variants", name + ".so")] + objs)
print(os.path.join(CSRC, "variants", name + ".so"))
