"""experiment: one literal ply on a one-board handle -- launch call, data visible in host-pinned memory, stream wait"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hex_gym_env_amd._engine import _BoardEngine
from hex_gym_env_amd import _lib

for n in (5, 11):
    e = _BoardEngine(n, "A")
    L, h = e._L, e._h
    tl, tv, tw, k = 0.0, 0.0, 0.0, 0
    rng = np.random.default_rng(0)
    for game in range(300):
        e.set_position(np.full((n, n), 2), None, 0)
        order = rng.permutation(n * n)
        for a in order[: n * n - 2]:
            e._act[0, 0] = a
            e._ret[0, 0] = 0x55
            t0 = time.perf_counter()
            L.hexvec_apply_moves(h, e._p_act, 1, e._p_ret, e._p_done, e._p_rb, e._p_lg, e._p_ob, e._st)
            t1 = time.perf_counter()
            ret = e._ret
            while ret[0, 0] == 0x55:
                pass
            t2 = time.perf_counter()
            L.hexvec_stream_synchronize(h, e._st)
            t3 = time.perf_counter()
            tl += t1 - t0; tv += t2 - t1; tw += t3 - t2; k += 1
            if e._done[0, 0]:
                break
    print("%dx%d: launch call %.2f us, result byte visible after %.2f us more, stream wait returns %.2f us after that (%d plies)"
          % (n, n, tl / k * 1e6, tv / k * 1e6, tw / k * 1e6, k))
    # the plain path for comparison
    t0 = time.perf_counter(); k2 = 0
    for game in range(300):
        e.set_position(np.full((n, n), 2), None, 0)
        for a in rng.permutation(n * n)[: n * n - 2]:
            r, d, _ = e.ply(int(a)); k2 += 1
            if d:
                break
    print("      ply() as shipped: %.2f us per ply (set_position included)" % ((time.perf_counter() - t0) / k2 * 1e6))
