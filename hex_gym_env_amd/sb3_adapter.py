"""HexSB3VecEnv: the stable-baselines3 `VecEnv` contract on top of HexVecEnv.

What sb3_contrib's MaskablePPO needs from a vectorised env: num_envs,
observation_space, action_space, reset() -> obs, step_async(actions), step_wait() ->
(obs, rewards, dones, infos) with auto reset, infos[i]["terminal_observation"] and
infos[i]["episode"] = {"r","l","t"} for finished envs, env_method("action_masks"),
has_attr("action_masks").  SB3 is optional: when it is importable this class derives
from stable_baselines3.common.vec_env.VecEnv, otherwise it is a plain class with the
same methods (the build image has no SB3).

The SB3 contract is numpy on the host; that round trip dominates at tens of thousands
of envs.  Use HexVecEnv directly (torch tensors on the device) for throughput.
"""
import time

import numpy as np
import torch

from ._compat import spaces
from .vec_env import HexVecEnv

try:  # pragma: no cover
    from stable_baselines3.common.vec_env import VecEnv as _Base
    HAVE_SB3 = True
except Exception:
    _Base = object
    HAVE_SB3 = False


class HexSB3VecEnv(_Base):
    def __init__(self, board_size=5, num_envs=64, agent_player_num=None, seed=0, device=None,
                 encoding="B", buffer_size=None, scores=None, base_model=None, sample_board=False):
        """buffer_size=None: the opponent is the library's uniform-legal policy (BaseRandomPolicy).
        buffer_size=k: the reference's opponent buffer (selfplay_wrapper kwargs buffer_size / scores /
        base_model / sample_board), with set_eval / get_scores / set_opponent_model ... as
        minihex/EvaluationCallback.py:21-56 calls them on `gym_env`."""
        self.sp = None
        if buffer_size is not None:
            if encoding != "B":
                raise ValueError("the opponent buffer is the SelfplayWrapper (encoding B)")
            from .vec_selfplay import VecSelfPlayEnv
            self.sp = VecSelfPlayEnv(board_size, num_envs, buffer_size=buffer_size, scores=scores,
                                     base_model=base_model, agent_player_num=agent_player_num, seed=seed,
                                     device=device, sample_board=sample_board)
            self.vec = self.sp.vec
        else:
            self.vec = HexVecEnv(board_size, num_envs, device=device, seed=seed, encoding=encoding,
                                 agent_player_num=agent_player_num)
        n = board_size
        lo, hi = (-1, 1) if encoding == "B" else (0, 2)
        obs_space = spaces.Box(low=lo, high=hi, shape=(n, n), dtype=np.int8)
        act_space = spaces.Discrete(n * n)
        if HAVE_SB3:
            super().__init__(num_envs, obs_space, act_space)
        else:
            self.num_envs, self.observation_space, self.action_space = num_envs, obs_space, act_space
        self._tobs = torch.zeros((num_envs, n, n), dtype=torch.int8, device=self.vec.device)
        # host side of a step: pinned staging buffers, filled by asynchronous copies behind the step's launches and
        # awaited ONCE (a `.cpu()` per array is a blocking pageable copy each: five stream waits per step)
        pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
        self._h_obs, self._h_legal = pin((num_envs, n, n), torch.int8), pin((num_envs, n * n), torch.uint8)
        self._h_rew, self._h_done = pin((num_envs,), torch.float32), pin((num_envs,), torch.uint8)
        self._h_tobs = pin((num_envs, n, n), torch.int8)
        self._actions = None
        self._ep_ret = np.zeros(num_envs, np.float64)
        self._ep_len = np.zeros(num_envs, np.int64)
        self._t0 = time.time()
        self._masks = None
        self._infos = [{} for _ in range(num_envs)]
        self.render_mode = None

    # -- VecEnv API ---------------------------------------------------------
    def reset(self):
        obs, legal = self.sp.reset() if self.sp is not None else self.vec.reset()
        self._h_obs.copy_(obs.view_as(self._h_obs), non_blocking=True)
        self._h_legal.copy_(legal, non_blocking=True)
        torch.cuda.current_stream(self.vec.device).synchronize()
        self._masks = self._h_legal.numpy().astype(bool)
        self._ep_ret[:] = 0
        self._ep_len[:] = 0
        return self._h_obs.numpy().copy()

    def step_async(self, actions):
        self._actions = np.asarray(actions)

    def step_wait(self):
        if self.sp is not None:
            obs, rew, done, legal, _ = self.sp.step(self._actions)
            self._tobs = self.sp.terminal_obs
        else:
            obs, rew, done, legal, _ = self.vec.step(self._actions, terminal_obs=self._tobs)
        self._h_obs.copy_(obs.view_as(self._h_obs), non_blocking=True)
        self._h_rew.copy_(rew, non_blocking=True)
        self._h_done.copy_(done, non_blocking=True)
        self._h_legal.copy_(legal, non_blocking=True)
        self._h_tobs.copy_(self._tobs, non_blocking=True)      # (small; saves a second wait on steps that end games)
        torch.cuda.current_stream(self.vec.device).synchronize()
        # fresh arrays: the caller may keep the previous step's (SB3 adds `_last_obs` to its buffer after stepping)
        obs_h, rew_h = self._h_obs.numpy().copy(), self._h_rew.numpy().copy()
        done_h = self._h_done.numpy().astype(bool)
        self._masks = self._h_legal.numpy().astype(bool)
        self._ep_ret += rew_h
        self._ep_len += 1
        for i in self._dirty:
            self._infos[i] = {}
        self._dirty = np.flatnonzero(done_h)
        if len(self._dirty):
            # one gather for all finished boards, then one dict per finished board (the VecEnv contract wants dicts)
            tobs = self._h_tobs.numpy()[self._dirty]
            now = time.time() - self._t0
            rets, lens = self._ep_ret[self._dirty].tolist(), self._ep_len[self._dirty].tolist()
            infos = self._infos
            for k, i in enumerate(self._dirty.tolist()):
                infos[i] = {"terminal_observation": tobs[k], "TimeLimit.truncated": False,
                            "episode": {"r": rets[k], "l": lens[k], "t": now}}
            self._ep_ret[self._dirty] = 0
            self._ep_len[self._dirty] = 0
        return obs_h, rew_h, done_h, self._infos

    _dirty = ()

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def action_masks(self):
        return self._masks

    def env_method(self, method_name, *args, indices=None, **kwargs):
        idx = self._indices(indices)
        if method_name == "action_masks":
            return [self._masks[i] for i in idx]
        raise AttributeError(f"HexSB3VecEnv has no per-env method {method_name!r}")

    def has_attr(self, attr_name):
        return attr_name in ("action_masks", "board_size")

    def get_attr(self, attr_name, indices=None):
        idx = self._indices(indices)
        if attr_name == "board_size":
            return [self.vec.board_size for _ in idx]
        if attr_name == "render_mode":
            return [None for _ in idx]
        raise AttributeError(attr_name)

    def set_attr(self, attr_name, value, indices=None):
        raise AttributeError("HexSB3VecEnv attributes are fixed at construction")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    # -- what SelfPlayCallback calls on gym_env (EvaluationCallback.py:31-54) -------------
    def _buffer(self):
        if self.sp is None:
            raise AttributeError("created without buffer_size: there is no opponent buffer")
        return self.sp

    def set_eval(self, eval_state):
        self._buffer().set_eval(eval_state)

    def get_scores(self):
        return self._buffer().get_scores()

    def get_opponent_models(self):
        return self._buffer().get_opponent_models()

    def set_opponent_model(self, index, model, score):
        self._buffer().set_opponent_model(index, model, score)

    def save_best_model(self):
        self._buffer().save_best_model()

    @property
    def best_score(self):
        return self._buffer().best_score

    def seed(self, seed=None):
        return [None] * self.num_envs

    def close(self):
        self.vec.close()

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices
