"""Single-board engine handle shared by the two HexGame look-alikes.

The reference's engines are plain numpy objects (HexGame.py:16-142,
HexSingleGame.py:21-153).  Here the state lives on the GPU in a one-env,
exact-label libhexvec handle; every `make_move` is one literal ply
(`hexvec_apply_moves`, T = 1), so the attributes the reference exposes
(.board, .regions, .done, .winner, .current_player_num) are always those of
the reference on the same move sequence.

A ply costs ONE kernel launch and ONE wait: the action and every output (return
code, done, reward, legal mask, board) live in host-pinned buffers that the
kernel reads and writes directly over the bus -- no device allocation, no copy
call; `hexvec_stream_synchronize` is the only blocking call.  regions and
current_player_num are fetched (same way) only when somebody asks for them.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .vec_env import HexVecEnv


def _pinned(shape, dtype):
    t = torch.empty(shape, dtype=dtype, pin_memory=True)
    return t, t.numpy(), C.c_void_p(t.data_ptr())


class _BoardEngine:
    """encoding 'A': engine plies; encoding 'B': env-B steps (make_move + inversion)."""

    def __init__(self, board_size, encoding, device=None):
        self.n = n = int(board_size)
        self.encoding = encoding
        self.env = HexVecEnv(n, 1, device=device, encoding=encoding, two_sided=True,
                             auto_reset=False)
        self._L, self._h = self.env._L, self.env._h
        cells, padded = n * n, (n + 2) ** 2
        # host-pinned I/O, written / read by the kernels themselves
        self._t_act, self._act, self._p_act = _pinned((1, 1), torch.int32)
        self._t_ret, self._ret, self._p_ret = _pinned((1, 1), torch.int8)
        self._t_done, self._done, self._p_done = _pinned((1, 1), torch.uint8)
        self._t_rb, self._rb, self._p_rb = _pinned((1, 1), torch.int8)
        self._t_lg, self._lg, self._p_lg = _pinned((cells,), torch.uint8)
        self._t_ob, self._ob, self._p_ob = _pinned((cells,), torch.int8)
        self._t_lab, self._lab, self._p_lab = _pinned((2 * padded,), torch.uint8)
        self._t_cur, self._curb, self._p_cur = _pinned((1,), torch.uint8)
        self._t_bin, self._bin, self._p_bin = _pinned((cells,), torch.int8)
        self._t_lin, self._lin, self._p_lin = _pinned((2 * padded,), torch.uint8)
        self._empty = 2 if encoding == "A" else 0
        self._board = self._legal = self._labels = self._cur = None
        self._st = self.env._stream()   # every call below ends in a wait, so one fixed stream is enough
        # completion ticket: the ply kernel adds 1 to this host-pinned word behind its outputs; ply() spins on it
        self._t_tk, self._tk, self._p_tk = _pinned((1,), torch.int32)
        self._tk = self._tk.view(np.uint32)     # the kernel counts modulo 2^32
        self._tk[0] = 0
        self._expect = 0
        _lib.check(self._L.hexvec_set_ticket(self._h, self._p_tk))
        # scalar I/O of a ply through ctypes views of the same pinned words (a numpy element access costs ~3x as much)
        self._c_act = C.c_int32.from_address(self._t_act.data_ptr())
        self._c_ret = C.c_int8.from_address(self._t_ret.data_ptr())
        self._c_done = C.c_uint8.from_address(self._t_done.data_ptr())
        self._c_rb = C.c_int8.from_address(self._t_rb.data_ptr())
        self._c_tk = C.c_uint32.from_address(self._t_tk.data_ptr())
        self._apply = self._L.hexvec_apply_moves
        self._apply_args = (self._h, self._p_act, 1, self._p_ret, self._p_done, self._p_rb, self._p_lg, self._p_ob,
                            self._st)
        self._fetch_board()

    # -- state transfer -------------------------------------------------
    def _wait(self):
        rc = self._L.hexvec_stream_synchronize(self._h, self._st)
        if rc:
            _lib.check(rc)
        self._expect = int(self._tk[0])    # launches that signalled before this wait are accounted for

    def _fetch_board(self):
        """board + labels + player to move of the current position (after a reset / set_position)"""
        _lib.check(self._L.hexvec_get_state(self._h, self._p_ob, self._p_lab, self._p_cur, self._st))
        self._wait()
        self._board = self._legal = None
        self._labels = self._lab.reshape(2, self.n + 2, self.n + 2).astype(np.float64)
        self._cur = int(self._curb[0])
        self._lg[:] = self._ob == self._empty

    def set_position(self, board, regions, cur):
        self._bin[:] = np.asarray(board).reshape(-1)
        lab = None
        if regions is not None:
            self._lin[:] = np.asarray(regions).reshape(-1)
            lab = self._p_lin
        _lib.check(self._L.hexvec_set_state(self._h, self._p_bin, lab, int(cur), self._st))
        self._fetch_board()

    def ply(self, action):
        """one literal ply; returns (ret code, done flag, reward[0])"""
        self._c_act.value = int(action)
        rc = self._apply(*self._apply_args)
        if rc:
            _lib.check(rc)
        # wait for the kernel's ticket (one board = +1 per launch) instead of a stream wait
        want = self._expect = (self._expect + 1) & 0xFFFFFFFF
        tk = self._c_tk
        for _ in range(400000):             # ~10 ms of polling, then the blocking wait decides
            if tk.value == want:
                break
        else:
            self._wait()
            if tk.value != want:
                raise RuntimeError("libhexvec: the ply kernel did not signal its completion ticket")
        self._board = self._legal = self._labels = self._cur = None
        return self._c_ret.value, self._c_done.value != 0, self._c_rb.value

    @property
    def board(self):
        if self._board is None:     # float64 like the reference's numpy boards (a fresh array per position)
            self._board = self._ob.reshape(self.n, self.n).astype(np.float64)
        return self._board

    @property
    def legal(self):
        if self._legal is None:
            self._legal = self._lg.astype(bool)
        return self._legal

    def _fetch_labels(self):
        _lib.check(self._L.hexvec_get_state(self._h, None, self._p_lab, self._p_cur, self._st))
        self._wait()
        self._labels = self._lab.reshape(2, self.n + 2, self.n + 2).astype(np.float64)
        self._cur = int(self._curb[0])

    @property
    def regions(self):
        if self._labels is None:
            self._fetch_labels()
        return self._labels

    @property
    def cur(self):
        if self._cur is None:
            self._fetch_labels()
        return self._cur


# Engines are pooled: the reference builds a fresh HexGame at every reset (HexGame.py:228, HexSingleGame.py:209), and
# a fresh engine here would mean a new device handle plus a dozen pinned allocations per game.  A HexGame look-alike
# takes an engine with acquire() and hands it back from __del__; set_position() re-initialises it completely.
# The pool keeps a strong reference to every engine it made (busy or not): an engine must never become part of a
# garbage cycle together with its HexGame, or the collector would run the device handle's finalizer before
# HexGame.__del__ could hand the engine back.
_POOL = {}
_POOL_MAX = 8


def acquire(board_size, encoding, device=None):
    engines = _POOL.setdefault((int(board_size), encoding, str(device)), [])
    for eng in engines:
        if not eng._busy:
            eng._busy = True
            eng._st = eng.env._stream()     # the NEW user's current stream (the first user's may be gone)
            return eng
    eng = _BoardEngine(board_size, encoding, device)
    eng._busy = True
    if len(engines) < _POOL_MAX:        # beyond that an engine simply lives and dies with its HexGame
        engines.append(eng)
    return eng


def release(eng):
    if eng is not None:
        eng._busy = False
