"""Single-board engine handle shared by the two HexGame look-alikes.

The reference's engines are plain numpy objects (HexGame.py:16-142,
HexSingleGame.py:21-153).  Here the state lives on the GPU in a one-env,
exact-label libhexvec handle; every `make_move` is one literal ply
(`hexvec_apply_moves`, T = 1) followed by a read-back, so the attributes the
reference exposes (.board, .regions, .done, .winner, .current_player_num)
are always those of the reference on the same move sequence.
"""
import numpy as np

from .vec_env import HexVecEnv


class _BoardEngine:
    """encoding 'A': engine plies; encoding 'B': env-B steps (make_move + inversion)."""

    def __init__(self, board_size, encoding, device=None):
        self.n = int(board_size)
        self.encoding = encoding
        self.env = HexVecEnv(self.n, 1, device=device, encoding=encoding, two_sided=True,
                             auto_reset=False)
        self._board = None
        self._legal = None
        self._sync()

    # -- state transfer -------------------------------------------------
    def _sync(self):
        board, labels, cur = self.env.get_state()
        self._board = board.cpu().numpy().reshape(self.n, self.n).astype(np.float64)
        self._labels = labels.cpu().numpy().reshape(2, self.n + 2, self.n + 2).astype(np.float64)
        self._cur = int(cur.cpu().numpy()[0])
        empty = 2 if self.encoding == "A" else 0
        self._legal = (self._board.ravel() == empty)

    def set_position(self, board, regions, cur):
        b = np.asarray(board).astype(np.int8).reshape(1, -1)
        lab = None if regions is None else np.asarray(regions).astype(np.uint8).reshape(1, -1)
        self.env.set_state(b, lab, cur)
        self._sync()

    def ply(self, action):
        """one literal ply; returns (ret code, done flag, reward[0])"""
        ret, done, rb, legal, obs = self.env.apply_moves(np.array([[int(action)]], np.int32))
        ret = int(ret.cpu().numpy()[0, 0])
        done = bool(done.cpu().numpy()[0, 0])
        rb = int(rb.cpu().numpy()[0, 0])
        self._board = obs.cpu().numpy().reshape(self.n, self.n).astype(np.float64)
        self._legal = legal.cpu().numpy().reshape(-1).astype(bool)
        self._labels = None
        self._cur = None
        return ret, done, rb

    @property
    def board(self):
        return self._board

    @property
    def legal(self):
        return self._legal

    @property
    def regions(self):
        if self._labels is None:
            self._sync()
        return self._labels

    @property
    def cur(self):
        if self._cur is None:
            self._sync()
        return self._cur
