"""Run the CPU oracle under AddressSanitizer + UBSan (host code only):

    make -C oracle asan
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_oracle.py
"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hex_oracle as ho
ho._SO = os.path.join(os.path.dirname(os.path.abspath(ho.__file__)), 'libhexoracle_asan.so')
ho.build = lambda force=False: ho._SO
import numpy as np
# golden tests through the ASan build
import tests.test_oracle_golden as t
for name in dir(t):
    if name.startswith('test_'):
        getattr(t, name)(); print('ok', name)
# vectorised layer: all modes, odd sizes
for flags in (ho.F_ENC_B|ho.F_AGENT_RANDOM, ho.F_ENC_B|ho.F_TWO_SIDED, 0, ho.F_TWO_SIDED|ho.F_FIRST_WHITE, ho.F_ENC_B|ho.F_AGENT_WHITE|ho.F_NO_AUTO_RESET):
    for n in (2, 5, 11, 19):
        o = ho.OracleVec(n, 37, seed=3, flags=flags)
        o.set_threads(3)
        if flags & ho.F_TWO_SIDED and not flags & ho.F_NO_AUTO_RESET:
            o.rollout(50)
            o.apply_moves(np.random.default_rng(0).integers(0, n*n, size=(20,37)).astype(np.int32))
        for _ in range(30): o.step()
        o.step(np.random.default_rng(1).integers(0, n*n, size=37).astype(np.int32))
        o.reset(env_mask=(np.arange(37)%3==0).astype(np.uint8)); o.get_state(); o.episode_stats()
print('vec layer ok')
