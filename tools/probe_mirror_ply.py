"""where a single-board mirror ply spends its time (run on the GPU box)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from hex_gym_env_amd._engine import _BoardEngine
from hex_gym_env_amd import _lib
e = _BoardEngine(11, "A")
L, h = e._L, e._h
st = e.env._stream()
def t(fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
k = [0]
def launch_only():
    e._act[0, 0] = -1          # illegal ply: state unchanged, same cost
    L.hexvec_apply_moves(h, e._p_act, 1, e._p_ret, e._p_done, e._p_rb, e._p_lg, e._p_ob, st)
def launch_sync():
    launch_only(); L.hexvec_stream_synchronize(h, st)
def launch_sync_torch():
    launch_only(); torch.cuda.current_stream().synchronize()
def full_ply():
    e.ply(-1)
def stream_call():
    e.env._stream()
print("launch only (async, queue fills) %.2f us" % t(launch_only))
print("launch + hexvec_stream_synchronize %.2f us" % t(launch_sync))
print("launch + torch stream.synchronize %.2f us" % t(launch_sync_torch))
print("engine.ply() %.2f us" % t(full_ply))
print("env._stream() %.2f us" % t(stream_call, 20000))
print("board property (astype float64) %.2f us" % t(lambda: e._ob.reshape(11, 11).astype(np.float64), 20000))
