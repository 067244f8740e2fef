"""HexVecEnv: tensor-native front end of libhexvec.

All observations / masks / rewards / dones are torch tensors on the GPU; the
calls enqueue on torch's current stream and never synchronise.  The semantics
per env are the reference's:

  self-play handle, encoding B  : SelfPlayEnv.step / reset  (minihex/SelfplayWrapper.py:69-89, :174-199)
  self-play handle, encoding A  : hex-v0 HexEnv.step / reset (minihex/HexGame.py:206-295)
  two-sided handle, encoding B  : HexEnv.step / reset        (minihex/HexSingleGame.py:208-263)
  two-sided handle, encoding A  : HexGame.make_move          (minihex/HexGame.py:85-111)

with VecEnv-style auto reset of finished envs (unless auto_reset=False).
"""
import ctypes as C

import torch

from . import _lib

__all__ = ["HexVecEnv"]


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


try:    # one C call instead of building a torch.cuda.Stream object per launch (~1.5 us of a ~7 us host-side step)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


class HexVecEnv:
    def __init__(self, board_size=11, num_envs=1024, device=None, seed=0, encoding="B",
                 two_sided=False, agent_player_num=None, auto_reset=True, first_player=0,
                 env_offset=0, exact_labels=False, rollout_layout=False):
        if not (_lib.MIN_BOARD <= board_size <= _lib.MAX_BOARD):
            raise ValueError(f"board_size must be in {_lib.MIN_BOARD}..{_lib.MAX_BOARD}")
        if board_size > _lib.MAX_BOARD_LABELS and (not auto_reset or exact_labels or rollout_layout is True):
            raise ValueError(f"board_size > {_lib.MAX_BOARD_LABELS} runs on the default layout of auto-reset "
                             "handles only (include/hexvec.h)")
        if encoding not in ("A", "B"):
            raise ValueError("encoding must be 'A' (hex-v0 / HexGame.py) or 'B' (HexSingleGame.py)")
        if not torch.cuda.is_available():
            raise RuntimeError("HexVecEnv needs an MI355X: libhexvec has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HexVecEnv runs on the GPU only (device must be cuda:N)")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self._dev_index = idx
        flags = 0
        if encoding == "B":
            flags |= _lib.F_ENC_B
            if agent_player_num is None:
                flags |= _lib.F_AGENT_RANDOM        # SelfplayWrapper.py:72-73
            elif agent_player_num == 1:
                flags |= _lib.F_AGENT_WHITE
            elif agent_player_num != 0:
                raise ValueError("agent_player_num must be None, 0 or 1")
        else:
            if agent_player_num not in (None, 0):
                raise NotImplementedError(
                    "hex-v0 with player_color=WHITE is broken in the reference (board and "
                    "regions desynchronise, HexGame.py:245-248); only the BLACK agent is supported")
            if first_player == 1:
                flags |= _lib.F_FIRST_WHITE
        if two_sided:
            flags |= _lib.F_TWO_SIDED
        if not auto_reset:
            flags |= _lib.F_NO_AUTO_RESET
        if exact_labels:
            flags |= _lib.F_EXACT_LABELS
        if rollout_layout == "auto":
            # The lane-per-board layout runs one playing wavefront per 64 boards and needs about one per
            # SIMD (1 024 on MI355X) to win; below ~3/4 of that the bitboard kernels (4x the wavefronts at
            # 9x9..11x11) fill the chip better (measured: DESIGN.md, occupancy table).
            rollout_layout = (num_envs + 63) // 64 >= 768 and board_size <= _lib.MAX_BOARD_LABELS
        if rollout_layout:      # state laid out for rollout_selfplay / rollout_random (include/hexvec.h)
            flags |= _lib.F_ROLLOUT_LAYOUT
        # auto-reset handles run on the bitboard / lane-per-board kernels (include/hexvec.h)
        self.exact = (not auto_reset) or exact_labels or board_size < 4
        self.board_size, self.num_envs, self.encoding = board_size, num_envs, encoding
        self.two_sided, self.auto_reset, self.flags = two_sided, auto_reset, flags
        self.cells = board_size * board_size
        self._L = _lib.lib()
        self._h = C.c_void_p()
        _lib.check(self._L.hexvec_create(C.byref(self._h), board_size, num_envs, idx,
                                         C.c_uint64(seed & (2 ** 64 - 1)), flags, env_offset))
        E, Cc, dev = num_envs, self.cells, self.device
        self.obs = torch.empty((E, board_size, board_size), dtype=torch.int8, device=dev)
        self.legal = torch.empty((E, Cc), dtype=torch.uint8, device=dev)
        self.reward = torch.empty(E, dtype=torch.float32, device=dev)
        self.done = torch.empty(E, dtype=torch.uint8, device=dev)
        self.winner = torch.empty(E, dtype=torch.int8, device=dev)
        self.pending = torch.zeros(E, dtype=torch.uint8, device=dev)
        # the env's own output buffers never move: their pointers are wrapped once (a host-stepped call is ~7 us of
        # Python around a ~5 us kernel at small env counts)
        self._po, self._pl, self._pr = _p(self.obs), _p(self.legal), _p(self.reward)
        self._pd, self._pw, self._pp = _p(self.done), _p(self.winner), _p(self.pending)

    # ------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.hexvec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        """the caller's current stream on this device, as the ABI wants it (queried per call: a graph capture or a
        `with torch.cuda.stream(...)` block changes it)"""
        return C.c_void_p(_raw_stream(self._dev_index))

    def _i32(self, a, shape):
        if a is None:
            return None
        a = torch.as_tensor(a, device=self.device)
        if a.dtype != torch.int32:
            a = a.to(torch.int32)
        a = a.contiguous()
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
        return a

    def _tobs(self, t):
        """a caller's terminal-observation buffer: int8 [E, N, N] on this device (written by the kernels)"""
        if t is not None and (t.dtype != torch.int8 or t.numel() != self.num_envs * self.cells
                              or t.device != self.device or not t.is_contiguous()):
            raise ValueError("terminal_obs must be a contiguous int8 [E, N, N] tensor on the env's device")
        return t

    # ------------------------------------------------------------------
    def reset(self, env_mask=None, opp_actions=None):
        """reset() of the masked envs (all by default) -> (obs int8 [E,N,N], legal uint8 [E,N*N])."""
        m = None
        if env_mask is not None:
            m = torch.as_tensor(env_mask, device=self.device).to(torch.uint8).contiguous()
            if m.shape != (self.num_envs,):
                raise ValueError("env_mask must have shape [num_envs]")
        o = self._i32(opp_actions, (self.num_envs,))
        _lib.check(self._L.hexvec_reset(self._h, _p(m), _p(o), _p(self.obs), _p(self.legal),
                                        self._stream()))
        return self.obs, self.legal

    def step(self, actions=None, opp_actions=None, terminal_obs=None):
        """One env step per board -> (obs, reward, done, legal, winner).

        actions None = uniform-legal agent drawn on the device (the random-policy
        workload); opp_actions None = uniform-legal opponent.  The returned
        tensors are the env's own output buffers (overwritten by the next call).
        """
        a = self._i32(actions, (self.num_envs,))
        o = self._i32(opp_actions, (self.num_envs,))
        self._tobs(terminal_obs)
        rc = self._L.hexvec_step(self._h, _p(a), _p(o), self._po, self._pl, self._pr, self._pd, self._pw,
                                 _p(terminal_obs), C.c_void_p(_raw_stream(self._dev_index)))
        if rc:
            _lib.check(rc)
        return self.obs, self.reward, self.done, self.legal, self.winner

    def _outs(self, out):
        """the five per-board output rows of a split-step call: the env's own buffers, or the caller's (`out`: any of
        obs int8 [E, N, N] / legal uint8 [E, N*N] / reward float32 [E] / done uint8 [E] / winner int8 [E]; e.g. row t of a
        rollout buffer, so that a collected step needs no copy).  -> (tensors, pointers)"""
        if not out:
            return (self.obs, self.legal, self.reward, self.done, self.winner), (self._po, self._pl, self._pr, self._pd, self._pw)
        E, Cc = self.num_envs, self.cells
        want = (("obs", torch.int8, E * Cc, self.obs), ("legal", torch.uint8, E * Cc, self.legal),
                ("reward", torch.float32, E, self.reward), ("done", torch.uint8, E, self.done),
                ("winner", torch.int8, E, self.winner))
        unknown = set(out) - {k for k, _, _, _ in want}
        if unknown:
            raise KeyError(f"unknown step output(s) {sorted(unknown)}")
        ts = []
        for k, dt, n, own in want:
            t = out.get(k)
            if t is None:
                t = own
            elif t.dtype != dt or t.device != self.device or not t.is_contiguous() or t.numel() != n:
                raise ValueError(f"step output {k!r} must be a contiguous {dt} tensor of {n} elements on {self.device}, "
                                 f"got {tuple(t.shape)} {t.dtype} on {t.device}")
            ts.append(t)
        return tuple(ts), tuple(_p(t) for t in ts)

    def step_agent(self, actions=None, terminal_obs=None, out=None):
        """Split step, first half: the agent's ply only -> (obs, legal, pending).

        Boards with pending == 1 show the OPPONENT's view; feed them to the opponent policy and
        call step_opponent() until pending is all zero (a second call plays the openings of boards
        that were reset with the opponent to move first).  reward / done / winner are complete
        after the last call.  out: see _outs()."""
        a = self._i32(actions, (self.num_envs,))
        self._tobs(terminal_obs)
        (obs, legal, _, _, _), (po, pl, pr, pd, pw) = self._outs(out)
        rc = self._L.hexvec_step_agent(self._h, _p(a), po, pl, pr, pd, pw,
                                       _p(terminal_obs), self._pp, C.c_void_p(_raw_stream(self._dev_index)))
        if rc:
            _lib.check(rc)
        return obs, legal, self.pending

    def step_opponent(self, opp_actions=None, terminal_obs=None, out=None):
        """Split step, second half: the opponent's ply on every pending board
        -> (obs, reward, done, legal, winner, pending).  out: see _outs()."""
        o = self._i32(opp_actions, (self.num_envs,))
        self._tobs(terminal_obs)
        (obs, legal, reward, done, winner), (po, pl, pr, pd, pw) = self._outs(out)
        rc = self._L.hexvec_step_opponent(self._h, _p(o), po, pl, pr, pd, pw,
                                          _p(terminal_obs), self._pp, C.c_void_p(_raw_stream(self._dev_index)))
        if rc:
            _lib.check(rc)
        return obs, reward, done, legal, winner, self.pending

    def apply_moves(self, actions):
        """Literal plies (no reset) from actions [T,E] -> ret, done, r_black, legal, obs (after each ply)."""
        a = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
        if a.dim() != 2 or a.shape[1] != self.num_envs:
            raise ValueError("actions must have shape [T, num_envs]")
        T, E, Cc, dev = a.shape[0], self.num_envs, self.cells, self.device
        ret = torch.empty((T, E), dtype=torch.int8, device=dev)
        done = torch.empty((T, E), dtype=torch.uint8, device=dev)
        rb = torch.empty((T, E), dtype=torch.int8, device=dev)
        legal = torch.empty((T, E, Cc), dtype=torch.uint8, device=dev)
        obs = torch.empty((T, E, Cc), dtype=torch.int8, device=dev)
        _lib.check(self._L.hexvec_apply_moves(self._h, _p(a), T, _p(ret), _p(done), _p(rb),
                                              _p(legal), _p(obs), self._stream()))
        return ret, done, rb, legal, obs

    def _rollout(self, fn, T, out):
        E, Cc, dev = self.num_envs, self.cells, self.device
        if out is None:
            out = dict(
                obs=torch.empty((T, E, Cc), dtype=torch.int8, device=dev),
                legal=torch.empty((T, E, Cc), dtype=torch.uint8, device=dev),
                actions=torch.empty((T, E), dtype=torch.int32, device=dev),
                reward=torch.empty((T, E), dtype=torch.float32, device=dev),
                done=torch.empty((T, E), dtype=torch.uint8, device=dev))
        # the kernels write T rows into whatever they are handed: a short, misplaced or mistyped buffer must raise
        # here, not fault the GPU
        want = {"obs": ((T, E, Cc), torch.int8), "legal": ((T, E, Cc), torch.uint8), "actions": ((T, E), torch.int32),
                "reward": ((T, E), torch.float32), "done": ((T, E), torch.uint8)}
        for k, t in out.items():
            if t is None:
                continue
            if k not in want:
                raise KeyError(f"unknown rollout output {k!r}")
            shape, dt = want[k]
            if (t.dtype != dt or t.device != dev or not t.is_contiguous() or t.numel() < T * E * (Cc if len(shape) == 3 else 1)):
                raise ValueError(f"rollout output {k!r} must be a contiguous {dt} tensor on {dev} with at least "
                                 f"{shape} elements, got {tuple(t.shape)} {t.dtype} on {t.device}")
        _lib.check(fn(self._h, T, _p(out.get("obs")), _p(out.get("legal")), _p(out.get("actions")),
                      _p(out.get("reward")), _p(out.get("done")), self._stream()))
        return out

    def rollout_random(self, T, out=None):
        """T plies per board, both sides uniform-legal, state in registers (two-sided handles)."""
        return self._rollout(self._L.hexvec_rollout_random, T, out)

    def rollout_selfplay(self, T, out=None):
        """T whole env steps per board with RNG agent and RNG opponent (self-play handles)."""
        return self._rollout(self._L.hexvec_rollout_selfplay, T, out)

    def get_state(self):
        """(board int8 [E,N*N], labels uint8 [E,2,(N+2)^2] or None, cur uint8 [E]).

        labels (the reference's `regions`) exist only on exact handles
        (auto_reset=False or exact_labels=True)."""
        E, P, dev = self.num_envs, (self.board_size + 2) ** 2, self.device
        board = torch.empty((E, self.cells), dtype=torch.int8, device=dev)
        labels = torch.empty((E, 2, P), dtype=torch.uint8, device=dev) if self.exact else None
        cur = torch.empty(E, dtype=torch.uint8, device=dev)
        _lib.check(self._L.hexvec_get_state(self._h, _p(board), _p(labels), _p(cur), self._stream()))
        return board, labels, cur

    def set_state(self, board, labels=None, cur_player=0):
        """HexGame.__init__ on given boards (exact handles): board int8 [E,N*N] in this handle's
        encoding; labels uint8 [E,2,(N+2)^2] adopted as connected_stones, or None to rebuild."""
        b = torch.as_tensor(board, device=self.device).to(torch.int8).contiguous()
        if b.numel() != self.num_envs * self.cells:
            raise ValueError("board must have E*N*N elements")
        l = None
        if labels is not None:
            l = torch.as_tensor(labels, device=self.device).to(torch.uint8).contiguous()
            if l.numel() != self.num_envs * 2 * (self.board_size + 2) ** 2:
                raise ValueError("labels must have E*2*(N+2)^2 elements")
        _lib.check(self._L.hexvec_set_state(self._h, _p(b), _p(l), int(cur_player), self._stream()))

    def set_boards(self, boards, env_mask=None):
        """start positions for the masked boards (auto-reset handles): boards int8 [E,N,N] in the
        frame of BLACK to move -> (obs, legal, pending)"""
        b = torch.as_tensor(boards, device=self.device).to(torch.int8).contiguous()
        if b.numel() != self.num_envs * self.cells:
            raise ValueError("boards must have E*N*N elements")
        m = None
        if env_mask is not None:
            m = torch.as_tensor(env_mask, device=self.device).to(torch.uint8).contiguous()
            if m.shape != (self.num_envs,):
                raise ValueError("env_mask must have shape [num_envs]")
        _lib.check(self._L.hexvec_set_boards(self._h, _p(m), _p(b), _p(self.obs), _p(self.legal),
                                             _p(self.pending), self._stream()))
        return self.obs, self.legal, self.pending

    def state_dict(self):
        """checkpoint of the env state (boards, connectivity, players, RNG counters, episode
        accumulators) as one opaque uint8 tensor plus the arguments it is valid for"""
        blob = torch.empty(int(self._L.hexvec_state_bytes(self._h)), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.hexvec_export_state(self._h, _p(blob), self._stream()))
        return {"blob": blob, "board_size": self.board_size, "num_envs": self.num_envs,
                "flags": self.flags}

    def load_state_dict(self, sd):
        if (sd["board_size"], sd["num_envs"], sd["flags"]) != (self.board_size, self.num_envs, self.flags):
            raise ValueError("state_dict was taken from an env with different board_size/num_envs/flags")
        blob = sd["blob"].to(self.device).contiguous()
        if blob.numel() != int(self._L.hexvec_state_bytes(self._h)):
            raise ValueError("state blob has the wrong size")
        _lib.check(self._L.hexvec_import_state(self._h, _p(blob), self._stream()))

    def render(self, indices=(0,)):
        """ASCII boards of the selected envs in the reference's render() layout (HexGame.py:305-330)"""
        from .render import board_to_text
        board = self.get_state()[0].cpu().numpy().reshape(self.num_envs, self.board_size, self.board_size)
        empty, black = (2, 0) if self.encoding == "A" else (0, -1)
        return "\n".join(board_to_text(board[i], empty, black) for i in indices)

    def episode_stats(self):
        """float64 [4] on the device: sum_return, sum_len, n_episodes, n_agent_wins."""
        out = torch.empty(4, dtype=torch.float64, device=self.device)
        _lib.check(self._L.hexvec_episode_stats(self._h, _p(out), self._stream()))
        return out

    def agent_colours(self):
        out = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        _lib.check(self._L.hexvec_agent_colours(self._h, _p(out), self._stream()))
        return out

    def last_kernel(self):
        return self._L.hexvec_last_kernel().decode()
