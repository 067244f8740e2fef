"""VecSelfPlayEnv: the SelfplayWrapper semantics for thousands of boards, opponents on the device.

The reference's SelfPlayEnv (minihex/SelfplayWrapper.py:37-208) keeps a buffer of opponent
policies, picks one per episode (80 % the best model, 20 % uniformly from the buffer,
:97-104; in eval mode the buffer is walked in order, :92-96) and lets it answer every agent
ply.  Here every board has its own opponent slot; one env step is

    step_agent(actions)  ->  [ group pending boards by slot -> policy(obs, mask) ]  ->  step_opponent(...)

repeated while boards are pending (a second round plays the openings of boards that were reset
with the opponent to move first).  All tensors stay on the GPU.

Policies are objects with  choose_actions(obs int8 [B,N,N], legal uint8 [B,N*N]) -> int32 / int64 [B]
(the batched form of OpponentPolicy.choose_action(board, action_mask), :26-35):
RandomLegalPolicy (BaseRandomPolicy, :16-24) and MlpActorPolicy (an SB3 MaskablePPO "MlpPolicy"
actor, loadable from the reference's shipped checkpoints with weights_only=True).
"""
import collections
import io
import weakref
import zipfile

import numpy as np
import torch

from .opponent_buffer import OpponentBuffer
from .vec_env import HexVecEnv

__all__ = ["RandomLegalPolicy", "MlpActorPolicy", "MlpCritic", "ppo_minibatch_step", "PredictAdapter", "as_policy",
           "VecSelfPlayEnv", "sample_boards"]


def sample_boards(board_size, count, generator, device):
    """`count` random start positions with the distribution of HexEnv.random_board
    (HexSingleGame.py:300-331), drawn on the device: an m x l sub-rectangle (m, l uniform in
    [N//4, N-2]) at a uniform position, holding 2*floor(m*l*(0.5+0.5u)/2) stones, half -1 (BLACK)
    and half +1 (WHITE), uniformly arranged.  Returns int8 [count, N, N]."""
    n = board_size
    lo, hi = n // 4, n - 1                      # np.random.randint(n//4, n-1): [lo, hi)

    def randint(a, b):                          # uniform integers in [a, b), b may be a tensor
        u = torch.rand(count, generator=generator, device=device, dtype=torch.float64)
        return torch.minimum((u * (b - a)).long(), torch.as_tensor(b - a - 1, device=device)) + a

    m, l = randint(lo, hi), randint(lo, hi)
    r0, c0 = randint(0, n - m + 1), randint(0, n - l + 1)
    u = torch.rand(count, generator=generator, device=device, dtype=torch.float64)
    stones = ((m * l).double() * (0.5 + 0.5 * u)).floor().long() // 2 * 2
    yy = torch.arange(n, device=device).view(1, n, 1)
    xx = torch.arange(n, device=device).view(1, 1, n)
    inside = ((yy >= r0.view(-1, 1, 1)) & (yy < (r0 + m).view(-1, 1, 1))
              & (xx >= c0.view(-1, 1, 1)) & (xx < (c0 + l).view(-1, 1, 1))).reshape(count, n * n)
    keys = torch.rand((count, n * n), generator=generator, device=device)
    keys = torch.where(inside, keys, torch.full_like(keys, 2.0))      # cells outside sort last
    rank = keys.argsort(1).argsort(1)                                  # rank of every cell
    half = (stones // 2).view(-1, 1)
    board = torch.zeros((count, n * n), dtype=torch.int8, device=device)
    board[rank < half] = -1
    board[(rank >= half) & (rank < 2 * half)] = 1
    return board.view(count, n, n)


def sample_masked_torch(logits, legal, generator=None, bad_rows=None):
    """sample_masked() spelled in torch ops (eight small kernels): the distribution reference of the tests"""
    u = torch.rand(legal.shape, generator=generator, device=legal.device, dtype=torch.float32)
    g = -torch.log(-torch.log(u.clamp_(1e-20, 1.0 - 1e-7)))
    z = g if logits is None else logits.float() + g
    z = torch.where(legal != 0, z, torch.full_like(g, float("-inf")))
    best, idx = z.max(1)            # one kernel for value and index
    if bad_rows is not None:        # a row without a legal cell peaks at -inf, one with a NaN logit at NaN
        bad_rows.add_((~(best > float("-inf"))).sum())
    return idx


# draw counters of hexvec_sample_masked, one per (device, generator): (device, id(generator)) -> (generator, device
# int64[2] {draw counter, reserved}).  torch.Generator takes neither attributes nor weak references, so an entry HOLDS its
# generator: an id can then never be re-used by another generator while the entry exists (a fresh generator that inherits a
# dead one's counter would repeat or skip draws).  Oldest entries leave when the table is full.
_SAMPLER_STATE = collections.OrderedDict()
_SAMPLER_STATE_MAX = 1024


def _want(t, name, dtype, numel, device):
    """a tensor handed to libhexvec as a raw pointer: wrong dtype / size / device / layout must raise here, not turn
    into an out-of-bounds access on the GPU"""
    if t is None:
        return
    if t.dtype != dtype or t.device != device or not t.is_contiguous() or t.numel() < numel:
        raise ValueError(f"{name} must be a contiguous {dtype} tensor on {device} with at least {numel} elements, "
                         f"got {tuple(t.shape)} {t.dtype} on {t.device}")


_LDS_PER_WORKGROUP = 160 * 1024     # MI355X (the library refuses what exceeds the device's own figure)


def sample_masked(logits, legal, generator=None, bad_rows=None, row_active=None, out=None, rng=None):
    """One categorical sample per row from softmax(logits) restricted to legal != 0 (logits None = uniform over
    the legal cells; a [1, C] logits row is shared by all boards), by the Gumbel-max trick argmax(logits + G):
    ONE libhexvec call (hexvec_sample_masked, include/hexvec.h), no device-side assertion, capturable in a
    hipGraph -- the draw counter lives on the device and advances with every call, also under graph replay.
    `generator` only names the random stream (its initial seed keys the counter hash); torch's own stream is
    not consumed.  row_active (uint8 / bool [rows], optional): boards with 0 get action 0 and are skipped.
    out (int32 [rows], optional): where to write the actions.  rng = (state, seed, salt, advance), optional:
    draw from the caller's counter instead (several calls of one step share it under different salts and
    advance it once, with the last call).

    A DEGENERATE row -- no legal cell, or a NaN logit on a legal cell -- cannot be sampled; it yields
    action 0 (an illegal ply ends the episode with reward 0) and is COUNTED in `bad_rows` (int64 scalar on
    the device, incremented by the kernel) so that it never passes silently: the policies below own such a
    counter and VecSelfPlayEnv.degenerate_rows() reads them.  Returns int32 [rows]."""
    import ctypes as C
    from . import _lib
    if not legal.is_cuda:
        raise RuntimeError("sample_masked runs on the GPU only (libhexvec has no CPU path)")
    if legal.dim() != 2:
        raise ValueError("legal must be [rows, cells]")
    rows, cols = legal.shape
    dev = legal.device
    lg = legal if legal.dtype == torch.uint8 and legal.is_contiguous() else legal.to(torch.uint8).contiguous()
    lo, stride = None, 0
    if logits is not None:
        if logits.dim() != 2 or logits.shape[1] != cols or logits.shape[0] not in (1, rows) or logits.device != dev:
            raise ValueError(f"logits must be [rows or 1, {cols}] on {dev}, got {tuple(logits.shape)} on {logits.device}")
        if logits.shape[0] == 1 or logits.stride(0) == 0:   # one row for every board
            lo = logits[:1].float().contiguous()
        else:
            lo = logits if logits.dtype == torch.float32 and logits.is_contiguous() else logits.float().contiguous()
            stride = cols
    ra = None
    if row_active is not None:
        if row_active.numel() != rows or row_active.device != dev:
            raise ValueError("row_active must have one entry per row, on the device of legal")
        ra = row_active if row_active.dtype == torch.uint8 else row_active.to(torch.uint8)
        ra = ra.contiguous()
    st, seed, salt, advance = _draw_stream(legal.device, generator, rng)
    if out is None:
        out = torch.empty(rows, dtype=torch.int32, device=legal.device)
    _want(out, "out", torch.int32, rows, dev)
    _want(bad_rows, "bad_rows", torch.int64, 1, dev)
    _want(st, "the draw counter", torch.int64, 2, dev)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().hexvec_sample_masked(p(lo), stride, p(lg), p(ra), rows, cols, C.c_uint64(seed), salt,
                                               int(bool(advance)), p(st), p(out), p(bad_rows),
                                               C.c_void_p(torch.cuda.current_stream(legal.device).cuda_stream)))
    return out


def _draw_stream(device, generator, rng):
    """(state, seed, salt, advance) of a sampling call: the caller's counter, or the one named by `generator`"""
    if rng is not None:
        return rng
    key = (torch.device(device), id(generator))
    ent = _SAMPLER_STATE.get(key)
    if ent is None or ent[0] is not generator:
        ent = _SAMPLER_STATE[key] = (generator, torch.zeros(2, dtype=torch.int64, device=device))
        while len(_SAMPLER_STATE) > _SAMPLER_STATE_MAX:
            _SAMPLER_STATE.popitem(last=False)
    else:
        _SAMPLER_STATE.move_to_end(key)
    return ent[1], (generator.initial_seed() if generator is not None else 0x5EED5EED) & (2 ** 64 - 1), 0, 1


def mlp_actions(weights, cells, h1, h2, obs, legal, row_active=None, slot=None, slot_bias=0, policy_of_slot=None,
                deterministic=False, generator=None, rng=None, out=None, logits_out=None, bad_rows=None, scratch=None):
    """Actor forward + mask + sample for a batch of boards as ONE libhexvec call (hexvec_mlp_actions,
    include/hexvec.h): weights float32 [n_policies, hexvec_mlp_weights_floats] in the packed layout
    (MlpActorPolicy.pack_into); slot int64 [rows] (+ slot_bias, through policy_of_slot int32 if given) names each
    board's network; with several networks the boards are grouped by network on the device (scratch: uint8 tensor
    of hexvec_mlp_scratch_bytes, allocated here if None).  The draws are sample_masked()'s.  Returns int32 [rows]."""
    import ctypes as C
    from . import _lib
    if not legal.is_cuda:
        raise RuntimeError("mlp_actions runs on the GPU only (libhexvec has no CPU path)")
    rows, dev = legal.shape[0], legal.device
    if obs.device != dev or obs.numel() != rows * cells or legal.dim() != 2:
        raise ValueError("obs / legal do not describe the same %d-cell boards on one device" % cells)
    ob = obs.reshape(rows, -1)
    ob = ob if ob.dtype == torch.int8 and ob.is_contiguous() else ob.to(torch.int8).contiguous()
    lg = legal if legal.dtype == torch.uint8 and legal.is_contiguous() else legal.to(torch.uint8).contiguous()
    if ob.shape[1] != cells or lg.shape[1] != cells:
        raise ValueError("obs / legal do not have %d cells per board" % cells)
    ra = None
    if row_active is not None:
        if row_active.numel() != rows or row_active.device != dev:
            raise ValueError("row_active must have one entry per row, on the device of legal")
        ra = (row_active if row_active.dtype == torch.uint8 else row_active.to(torch.uint8)).contiguous()
    if weights.device != dev:
        raise ValueError("the packed weights must live on the device of the boards")
    weights = weights.view(-1, weights.shape[-1])
    n_pol = weights.shape[0]
    L = _lib.lib()
    if weights.dtype != torch.float32 or not weights.is_contiguous() or weights.shape[1] != L.hexvec_mlp_weights_floats(cells, h1, h2):
        raise ValueError("weights are not float32 [n_policies, hexvec_mlp_weights_floats(cells, h1, h2)]")
    n_slots = 1
    if slot is not None:
        _want(slot, "slot", torch.long, rows, dev)
        _want(policy_of_slot, "policy_of_slot", torch.int32, 1, dev)
        n_slots = n_pol if policy_of_slot is None else policy_of_slot.numel()
        if n_pol > 1 and scratch is None:
            scratch = torch.empty(L.hexvec_mlp_scratch_bytes(rows, n_pol), dtype=torch.uint8, device=legal.device)
    st, seed, salt, advance = _draw_stream(legal.device, generator, rng)
    if out is None:
        out = torch.empty(rows, dtype=torch.int32, device=legal.device)
    _want(out, "out", torch.int32, rows, dev)
    _want(logits_out, "logits_out", torch.float32, rows * cells, dev)
    _want(bad_rows, "bad_rows", torch.int64, 1, dev)
    _want(st, "the draw counter", torch.int64, 2, dev)
    if scratch is not None and (scratch.device != dev or scratch.dtype != torch.uint8 or not scratch.is_contiguous()):
        raise ValueError("scratch must be a contiguous uint8 tensor on the device of the boards")
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    _lib.check(L.hexvec_mlp_actions(p(ob), p(lg), p(ra), p(slot), slot_bias, p(policy_of_slot), n_slots, rows, cells,
                                    p(weights), n_pol, h1, h2, int(bool(deterministic)), C.c_uint64(seed), salt,
                                    int(bool(advance)), p(st), p(out), p(logits_out), p(bad_rows), p(scratch),
                                    0 if scratch is None else scratch.numel(),
                                    C.c_void_p(torch.cuda.current_stream(legal.device).cuda_stream)))
    return out


class _CountsBadRows:
    """per-policy device counter of degenerate sampling rows (see sample_masked); such policies also take
    choose_actions(obs, legal, row_active=..., out=...) (boards to skip / where to write)"""
    _bad_rows = None
    takes_row_active = True

    def bad_rows_counter(self, device):
        if self._bad_rows is None or self._bad_rows.device != torch.device(device):
            self._bad_rows = torch.zeros((), dtype=torch.long, device=device)
        return self._bad_rows


class RandomLegalPolicy(_CountsBadRows):
    """uniform over the empty cells, drawn with torch's generator on the device"""

    def __init__(self, generator=None):
        self.generator = generator

    def group_key(self):
        """policies with equal keys answer identically distributed and may share one batched call"""
        return ("random-legal", id(self.generator))

    def choose_actions(self, obs, legal, row_active=None, out=None, rng=None):
        return sample_masked(None, legal, self.generator, self.bad_rows_counter(legal.device), row_active, out, rng)

    def save_model(self, path):
        return None


class MlpActorPolicy(torch.nn.Module, _CountsBadRows):
    """The actor of sb3_contrib's MaskableActorCriticPolicy with the default MLP extractor:
    flatten -> float -> Linear-Tanh-Linear-Tanh (mlp_extractor.policy_net) -> Linear (action_net)
    -> logits, illegal actions masked out, categorical sample (or argmax)."""

    def __init__(self, n_cells, hidden=(64, 64), deterministic=False, generator=None):
        super().__init__()
        dims = [n_cells] + list(hidden)
        layers = []
        for a, b in zip(dims, dims[1:]):
            layers += [torch.nn.Linear(a, b), torch.nn.Tanh()]
        self.policy_net = torch.nn.Sequential(*layers)
        self.action_net = torch.nn.Linear(dims[-1], n_cells)
        self.deterministic = deterministic
        self.generator = generator
        self.n_cells, self.hidden = n_cells, tuple(hidden)
        # the fused libhexvec call serves the SB3-default shape family; anything else keeps the torch spelling
        self.fusable = len(self.hidden) == 2 and self._shape_fits(n_cells, self.hidden)
        self._packed = self._packed_version = None
        self._compute_device = None     # None: where the parameters live; view_of_sb3() names the env's GPU
        self._owner = None

    @staticmethod
    def _shape_fits(n_cells, hidden):
        """can hexvec_mlp_actions serve this network: widths multiples of 16 in [16, 256] AND the kernel's LDS need
        within what one workgroup gets (256-256 at 16x16 and larger does not fit: the torch forward is used)"""
        if not all(h % 16 == 0 and 16 <= h <= 256 for h in hidden):
            return False
        from . import _lib
        need = int(_lib.lib().hexvec_mlp_lds_bytes(int(n_cells), int(hidden[0]), int(hidden[1])))
        return 0 < need <= _LDS_PER_WORKGROUP

    @classmethod
    def view_of_sb3(cls, model, device, deterministic=False, generator=None):
        """LIVE view of the actor of an SB3 / sb3_contrib model with an MLP policy (what the reference's
        OpponentPolicy wraps, SelfplayWrapper.py:26-35): the same nn.Linear objects, no copy -- the model may keep
        training and may live on the CPU; the packed weights on `device` follow its parameter versions.  Raises
        TypeError if the policy is not  flatten -> Linear-Tanh-Linear-Tanh -> Linear  (then predict() is used)."""
        pol = getattr(model, "policy", None)
        ext, act = getattr(pol, "mlp_extractor", None), getattr(pol, "action_net", None)
        pnet = getattr(ext, "policy_net", None)
        if pnet is None or not isinstance(act, torch.nn.Linear):
            raise TypeError("not an actor-critic MLP policy")
        fe = getattr(pol, "features_extractor", None)
        if fe is not None and type(fe).__name__ not in ("FlattenExtractor", "Flatten"):
            raise TypeError("the policy has a feature extractor other than flatten")
        mods = list(pnet)
        lins = [m for m in mods if isinstance(m, torch.nn.Linear)]
        if len(lins) != 2 or len(mods) != 4 or not all(isinstance(m, torch.nn.Tanh) for m in mods[1::2]):
            raise TypeError("the actor is not Linear-Tanh-Linear-Tanh")
        if act.in_features != lins[1].out_features or act.out_features != lins[0].in_features:
            raise TypeError("the actor does not map the board to one logit per cell")
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self.policy_net, self.action_net = pnet, act            # shared modules: live weights
        self.deterministic, self.generator = deterministic, generator
        self.n_cells, self.hidden = int(lins[0].in_features), (int(lins[0].out_features), int(lins[1].out_features))
        self.fusable = self._shape_fits(self.n_cells, self.hidden)
        self._packed = self._packed_version = None
        self._compute_device = torch.device(device)
        self._owner = weakref.ref(model)        # (weak: the view is cached per model, see as_policy)
        return self

    @property
    def compute_device(self):
        return self._compute_device if self._compute_device is not None else self.action_net.weight.device

    @classmethod
    def from_state_dict(cls, sd, **kw):
        ws = sorted(k for k in sd if k.startswith("mlp_extractor.policy_net.") and k.endswith(".weight"))
        hidden = [int(sd[k].shape[0]) for k in ws]
        pol = cls(int(sd[ws[0]].shape[1]), hidden, **kw)
        lin = [m for m in pol.policy_net if isinstance(m, torch.nn.Linear)]
        with torch.no_grad():
            for m, k in zip(lin, ws):
                m.weight.copy_(torch.as_tensor(sd[k]))
                m.bias.copy_(torch.as_tensor(sd[k[:-6] + "bias"]))
            pol.action_net.weight.copy_(torch.as_tensor(sd["action_net.weight"]))
            pol.action_net.bias.copy_(torch.as_tensor(sd["action_net.bias"]))
        return pol

    @classmethod
    def from_sb3_zip(cls, path, **kw):
        """policy.pth of an SB3 model zip; loaded with weights_only=True (nothing is executed)"""
        with zipfile.ZipFile(path) as z:
            sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
        return cls.from_state_dict(sd, **kw)

    @classmethod
    def from_npz(cls, path, **kw):
        d = np.load(path)
        return cls.from_state_dict({k: d[k] for k in d.files}, **kw)

    @torch.no_grad()
    def logits(self, obs, legal):
        """masked logits by plain torch ops (fp32 reference of the fused kernel), on the device of obs"""
        x = obs.reshape(obs.shape[0], -1).float().to(self.action_net.weight.device)
        z = self.action_net(self.policy_net(x)).to(obs.device)
        return z.masked_fill(legal == 0, float("-inf"))

    @torch.no_grad()
    def masked_logits(self, obs_float, legal_bool):
        """as logits(), for inputs that are already float [B, C] / bool [B, C] on the parameters' device"""
        return self.action_net(self.policy_net(obs_float)).masked_fill(~legal_bool, float("-inf"))

    # -- the fused path: weights packed for hexvec_mlp_actions (include/hexvec.h) --
    def weights_version(self):
        """changes whenever a parameter is written in place (optimizer step, load_state_dict) or replaced (.to())"""
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())   # (the module's structure is fixed)
        return tuple([(p.data_ptr(), p._version) for p in plist])

    def packed_floats(self):
        from . import _lib
        return int(_lib.lib().hexvec_mlp_weights_floats(self.n_cells, *self.hidden))

    @torch.no_grad()
    def pack_into(self, dst):
        """write this network into dst (float32 [packed_floats()], device memory) in the library's packed form
        (hexvec_mlp_pack: one small kernel; the layout is the library's business)"""
        import ctypes as C
        from . import _lib
        l1, l2 = [m for m in self.policy_net if isinstance(m, torch.nn.Linear)]
        src = [t.detach() for t in (l1.weight, l1.bias, l2.weight, l2.bias, self.action_net.weight, self.action_net.bias)]
        if not dst.is_cuda:
            raise RuntimeError("pack_into: dst must live on a GPU (libhexvec has no CPU path)")
        # (a model that lives on the CPU or another GPU is copied over: 6 small tensors, only when it changed)
        src = [t.to(device=dst.device, dtype=torch.float32).contiguous() for t in src]
        if dst.dtype != torch.float32 or not dst.is_contiguous() or dst.numel() != self.packed_floats():
            raise ValueError("dst is not a contiguous float32 tensor of packed_floats() elements")
        _lib.check(_lib.lib().hexvec_mlp_pack(*[C.c_void_p(t.data_ptr()) for t in src], self.n_cells, self.hidden[0],
                                              self.hidden[1], C.c_void_p(dst.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream(dst.device).cuda_stream)))
        return dst

    def packed(self):
        """the packed weights on the compute device, re-packed when the parameters changed"""
        v = self.weights_version()
        if self._packed is None or self._packed.device != self.compute_device:
            self._packed = torch.empty(self.packed_floats(), dtype=torch.float32, device=self.compute_device)
            self._packed_version = None
        if self._packed_version != v:
            self.pack_into(self._packed)
            self._packed_version = v
        return self._packed

    @torch.no_grad()
    def opening_logits(self):
        """logits of the EMPTY board, float32 [1, n_cells] (every cell legal): the opening distribution of this
        network.  Cached in a static buffer and recomputed only when the parameters changed -- a self-play step
        samples openings from it without running the network."""
        v = self.weights_version()
        dev, pdev = self.compute_device, self.action_net.weight.device
        if getattr(self, "_open_logits", None) is None or self._open_logits.device != dev:
            self._open_logits = torch.empty((1, self.n_cells), dtype=torch.float32, device=dev)
            self._open_version = None
        if self._open_version != v:     # (evaluated where the parameters live, kept on the compute device)
            self._open_logits.copy_(self.action_net(self.policy_net(torch.zeros((1, self.n_cells), device=pdev))))
            self._open_version = v
        return self._open_logits

    def sync_weights(self):
        """refresh everything derived from the parameters (call before replaying a captured graph that
        contains choose_actions / openings of this policy)"""
        self.opening_logits()
        return self.packed() if self.fusable else None

    @torch.no_grad()
    def choose_actions(self, obs, legal, row_active=None, out=None, rng=None):
        if self.fusable:         # forward + mask + sample (or arg-max): one launch
            return mlp_actions(self.packed(), self.n_cells, self.hidden[0], self.hidden[1], obs, legal, row_active,
                               deterministic=self.deterministic, generator=self.generator, rng=rng, out=out,
                               bad_rows=self.bad_rows_counter(legal.device))
        if self.deterministic:
            a = self.logits(obs, legal).argmax(1).to(torch.int32)
            return a if out is None else out.copy_(a)
        x = obs.reshape(obs.shape[0], -1).float().to(self.action_net.weight.device)
        return sample_masked(self.action_net(self.policy_net(x)).to(legal.device), legal, self.generator,
                             self.bad_rows_counter(legal.device), row_active, out, rng)

    def save_model(self, path):
        owner = self._owner() if self._owner is not None else None
        if owner is not None:           # a view of an SB3 model: OpponentPolicy.save_model (SelfplayWrapper.py:37-38)
            return owner.save(path)
        torch.save(self.state_dict(), path)


class MlpCritic(torch.nn.Module):
    """The critic of SB3's default MlpPolicy: flatten -> Linear-Tanh-Linear-Tanh (mlp_extractor.value_net) ->
    Linear(., 1) (value_net).  Training-side only: no kernel of libhexvec touches it."""

    def __init__(self, n_cells, hidden=(64, 64)):
        super().__init__()
        dims = [n_cells] + list(hidden)
        layers = []
        for a, b in zip(dims, dims[1:]):
            layers += [torch.nn.Linear(a, b), torch.nn.Tanh()]
        self.value_net = torch.nn.Sequential(*layers, torch.nn.Linear(dims[-1], 1))

    def forward(self, obs):
        return self.value_net(obs.reshape(obs.shape[0], -1).float()).squeeze(-1)


def ppo_minibatch_step(actor, critic, optimizer, obs, legal, actions, old_logp, advantages, returns,
                       clip=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, allreduce=False):
    """One optimizer step of MaskablePPO's update on one minibatch (the loss of sb3_contrib's train(): clipped
    surrogate on the masked categorical, value MSE, entropy bonus, gradient clipping) in plain fp32 torch.  The
    reference's scripts run this inside sb3_contrib (scripts/experiments/*.py:38-47); here it exists so that the
    update leg of a training iteration can be timed next to the collection step and so that
    dist.allreduce_gradients has a caller (allreduce=True: data-parallel step, one flattened all-reduce).
    obs float [B, C], legal bool [B, C], actions int64 [B], old_logp / advantages / returns float [B].
    Returns the loss (a device scalar; nothing synchronises)."""
    logits = actor.action_net(actor.policy_net(obs)).masked_fill(~legal, float("-inf"))
    logp_all = torch.log_softmax(logits, -1)
    logp = logp_all.gather(1, actions[:, None]).squeeze(1)
    ratio = torch.exp(logp - old_logp)
    adv = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    pg = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1.0 - clip, 1.0 + clip)).mean()
    vf = torch.nn.functional.mse_loss(critic(obs), returns)
    p_all = logp_all.exp()
    ent = -(p_all * logp_all.masked_fill(~legal, 0.0)).sum(-1).mean()
    loss = pg + vf_coef * vf - ent_coef * ent
    optimizer.zero_grad(set_to_none=False)
    loss.backward()
    params = [p for g in optimizer.param_groups for p in g["params"]]
    if allreduce:
        from . import dist as hxdist
        hxdist.allreduce_gradients(params)
    torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
    optimizer.step()
    return loss.detach()


class PredictAdapter:
    """Any model with the SB3 contract  predict(obs, deterministic=..., action_masks=...) -> (actions, _)
    (what the reference's OpponentPolicy wraps, SelfplayWrapper.py:26-35), called once per group of
    boards with batched numpy inputs.  The model object is kept by reference, like the reference does
    (set_opponent_model stores the live model, :125-136)."""

    def __init__(self, model, deterministic=False):
        self.opponent_model = model
        self.deterministic = deterministic

    def choose_actions(self, obs, legal):
        a, _ = self.opponent_model.predict(obs.cpu().numpy(), deterministic=self.deterministic,
                                           action_masks=legal.cpu().numpy().astype(bool))
        return torch.as_tensor(np.asarray(a), device=obs.device).long()

    def save_model(self, path):
        self.opponent_model.save(path)


_LIVE_VIEWS = weakref.WeakKeyDictionary()     # SB3-style model -> {device: MlpActorPolicy view}


def as_policy(model, device=None):
    """device policy for whatever set_opponent_model() is given: a policy of this module as is; an SB3-style model
    whose actor is the default MLP as a LIVE view on the fused kernel (MlpActorPolicy.view_of_sb3); any other model
    with predict() through PredictAdapter (batched numpy calls on the host)"""
    if hasattr(model, "choose_actions"):
        return model
    if hasattr(model, "predict"):
        if device is not None and torch.device(device).type == "cuda":
            # one view per (model, device): the reference's callback hands the SAME live model to slot after slot
            # (EvaluationCallback.py:45-49), and slots that share a policy object share a network in the pool table
            # (kept beside the model, not on it: an SB3 model serialises its __dict__ when it is saved)
            try:
                views = _LIVE_VIEWS.setdefault(model, {})
            except TypeError:            # a model that cannot be weakly referenced: no sharing
                views = {}
            key = str(torch.device(device))
            if key in views:
                return views[key]
            try:
                views[key] = MlpActorPolicy.view_of_sb3(model, device)
                return views[key]
            except TypeError:
                pass
        return PredictAdapter(model)
    raise TypeError("an opponent needs choose_actions(obs, legal) or predict(obs, deterministic, action_masks)")


class _PoolTable:
    """The pool's actor networks as ONE weights table for hexvec_mlp_actions: entry 0 of policy_of_slot is the best
    model, entry k + 1 pool member k; slots holding the same policy object share a network.  sync() re-packs the
    networks whose parameters changed (the reference keeps live models in the pool, SelfplayWrapper.py:125-136)."""

    def __init__(self, by_slot, device, num_envs):
        from . import _lib
        uniq, pos = {}, []
        for pol in by_slot:
            pos.append(uniq.setdefault(id(pol), (len(uniq), pol))[0])
        self.policies = [pol for _, pol in sorted(uniq.values(), key=lambda t: t[0])]
        first = self.policies[0]
        self.cells, self.hidden, self.deterministic = first.n_cells, first.hidden, bool(first.deterministic)
        self.table = torch.empty((len(self.policies), first.packed_floats()), dtype=torch.float32, device=device)
        self.policy_of_slot = torch.tensor(pos, dtype=torch.int32, device=device)
        self.versions = [None] * len(self.policies)
        self.scratch = None
        if len(self.policies) > 1:
            self.scratch = torch.empty(_lib.lib().hexvec_mlp_scratch_bytes(num_envs, len(self.policies)),
                                       dtype=torch.uint8, device=device)
        self.bad_rows = first.bad_rows_counter(device)
        self.sync()

    @staticmethod
    def fits(by_slot, device):
        first = by_slot[0]
        return all(isinstance(p, MlpActorPolicy) and p.fusable and p.n_cells == first.n_cells
                   and p.hidden == first.hidden and bool(p.deterministic) == bool(first.deterministic)
                   and p.compute_device == torch.device(device) for p in by_slot)

    def sync(self):
        for u, pol in enumerate(self.policies):
            v = pol.weights_version()
            if v != self.versions[u]:
                pol.pack_into(self.table[u])
                self.versions[u] = v

    def actions(self, obs, legal, slot, row_active, out, generator, rng):
        self.sync()
        return mlp_actions(self.table, self.cells, self.hidden[0], self.hidden[1], obs, legal, row_active,
                           slot=slot, slot_bias=1, policy_of_slot=self.policy_of_slot,
                           deterministic=self.deterministic, generator=generator, rng=rng, out=out,
                           bad_rows=self.bad_rows, scratch=self.scratch)


class VecSelfPlayEnv:
    """selfplay_wrapper(HexEnv)(board_size, buffer_size, scores, ...) for num_envs boards."""

    def __init__(self, board_size=5, num_envs=1024, buffer_size=20, scores=None, base_model=None,
                 agent_player_num=None, seed=0, device=None, prob_best=0.8, sample_board=False):
        self.vec = HexVecEnv(board_size, num_envs, device=device, seed=seed, encoding="B",
                             agent_player_num=agent_player_num)
        self.device, self.num_envs, self.board_size = self.vec.device, num_envs, board_size
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(seed) + 0x5EED)
        if base_model is None:           # SelfplayWrapper.py:58-60: random pool, zero scores
            pool = [RandomLegalPolicy(self._gen) for _ in range(buffer_size)]
            self.buffer = OpponentBuffer(pool, np.zeros(buffer_size), RandomLegalPolicy(self._gen), prob_best)
        else:                            # :54-57: the given model in every slot
            base_model = as_policy(base_model, self.device)
            pool = [base_model for _ in range(buffer_size)]
            self.buffer = OpponentBuffer(pool, np.zeros(buffer_size) if scores is None else scores,
                                         base_model, prob_best)
        self.sample_board = bool(sample_board)
        # opponent slot per board: -1 = best_model, k >= 0 = buffer slot k
        self.slot = torch.full((num_envs,), -1, dtype=torch.long, device=self.device)
        self._opp = torch.zeros(num_envs, dtype=torch.int32, device=self.device)
        self.terminal_obs = torch.zeros((num_envs, board_size, board_size), dtype=torch.int8,
                                        device=self.device)
        self._white = self.vec.agent_colours() != 0       # fixed per board (SelfplayWrapper.py:72-73)
        self._any_white = bool(self._white.any())          # one host read, at construction
        self._eval_cursor = torch.zeros((), dtype=torch.long, device=self.device)
        # hexvec_pick_opponents: the boards' "ended on the agent's ply" bytes, the agents' colours as bytes,
        # the draw counter of the slot choice
        self._ended = torch.zeros(num_envs, dtype=torch.uint8, device=self.device)
        self._white_u8 = self._white.to(torch.uint8)
        self._pick_state = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._pick_seed = (int(seed) * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & (2 ** 64 - 1)
        self._empty_board = torch.zeros((num_envs, board_size, board_size), dtype=torch.int8, device=self.device)
        self._all_legal = torch.ones((num_envs, board_size * board_size), dtype=torch.uint8, device=self.device)
        # hexvec_attach_pool: the slot choice and the openings ride in the tails of the two env launches of a step
        # (fuse_pool = False keeps them as launches of their own: the form the fused one is tested against)
        self.fuse_pool = True
        self._attached = None           # fingerprint of what the handle currently holds
        self._open_cdf = None           # float32 [n_policies, cells]: static buffer (captured graphs read it)
        self._open_cdf_versions = None
        self._open_bad = torch.zeros((), dtype=torch.long, device=self.device)

    # the reference's attribute names
    opponent_models = property(lambda self: self.buffer.members)
    opponent_scores = property(lambda self: self.buffer.scores)
    best_model = property(lambda self: self.buffer.best)
    best_score = property(lambda self: self.buffer.best_score)
    best_mean_reward = property(lambda self: self.buffer.best_mean_reward)
    eval_state = property(lambda self: self.buffer.evaluating)

    # -- opponent selection (setup_opponents, :91-104), one slot per board ------------------
    # Everything below is mask based and never reads a device value on the host: no
    # synchronisation inside step().
    def _draw_candidates(self):
        """one candidate slot per board for this step (a board ends at most once per step)"""
        E = self.num_envs
        u = torch.rand(E, generator=self._gen, device=self.device)
        k = torch.randint(len(self.buffer), (E,), generator=self._gen, device=self.device)
        self._cand = torch.where(u < self.buffer.prob_best, torch.full_like(k, -1), k)

    def _setup_opponents(self, mask):
        """new opponent slot for the boards in `mask` (bool [E]); the others keep theirs"""
        if self.buffer.evaluating:       # walk the pool in order, one member per new episode
            rank = torch.cumsum(mask.to(torch.long), 0) - 1
            new = (rank + self._eval_cursor) % len(self.buffer)
            self._eval_cursor.add_(mask.sum())                      # device-side cursor, in place
        else:
            new = self._cand
        self.slot.copy_(torch.where(mask, new, self.slot))         # in place: hipGraph replays read this tensor

    def _groups(self):
        """slots grouped by policy: the same OBJECT (the reference's pool mostly holds the same live
        model) or an equal group_key() (stateless policies such as the default random pool)"""
        groups = {}
        for k in range(-1, len(self.buffer)):
            pol = self.buffer.policy(k)
            key = pol.group_key() if hasattr(pol, "group_key") else id(pol)
            groups.setdefault(key, (pol, []))[1].append(k)
        return groups

    def _pool_table(self):
        """the pool as one weights table if every member is an MlpActorPolicy of one shape on this device
        (then an opponent round is ONE policy launch whatever the pool holds), else None"""
        key = (id(self.buffer), self.buffer.generation)     # (the buffer outlives the key; ids of its members may not)
        if getattr(self, "_table_key", None) != key:
            by_slot = [self.buffer.policy(k) for k in range(-1, len(self.buffer))]
            self._table = _PoolTable(by_slot, self.device, self.num_envs) if _PoolTable.fits(by_slot, self.device) else None
            self._table_key = key
        return self._table

    def _rng(self, salt, advance=False):
        """the step's shared draw counter for the fused calls: distinct salts, advanced once (by the last)"""
        return (self._pick_state, self._pick_seed, salt, advance)

    def _pick(self, when, salt=0, advance=False):
        """setup_opponents (:97-104) for the boards that start a new episode at point `when` of the step, as
        ONE libhexvec launch (hexvec_pick_opponents: the selection mask and the 80/20 draw are fused); the
        evaluation walk (:92-96) needs a running count over the boards and keeps the torch spelling"""
        import ctypes as C
        from . import _lib
        v = self.vec
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(_lib.lib().hexvec_pick_opponents(
            p(self.slot), when, p(v.done), p(v.pending), p(self._ended), p(self._white_u8), self.num_envs,
            C.c_float(self.buffer.prob_best), len(self.buffer), C.c_uint64(self._pick_seed), salt, int(advance),
            p(self._pick_state),
            C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _opponent_actions(self, obs, legal, pending, rng=None):
        """actions of every pending board's opponent: each distinct policy group is evaluated on the
        whole batch and its answers are kept where that policy is the opponent."""
        tab = self._pool_table()
        if tab is not None:
            tab.actions(obs, legal, self.slot, pending, self._opp, self._gen, rng)
            return
        groups = self._groups()
        safe_legal = None
        first = True
        for pol, slots in groups.values():
            if getattr(pol, "takes_row_active", False) and len(groups) == 1:      # the sampler skips the boards that owe no ply and writes in place
                pol.choose_actions(obs, legal, row_active=pending, out=self._opp, rng=rng)
                return
            if safe_legal is None:              # generic policies: boards that owe no ply look all-legal to them
                safe_legal = torch.where(pending.view(-1, 1).bool(), legal, torch.ones_like(legal))
            a = pol.choose_actions(obs, safe_legal).to(torch.int32)
            if first and len(groups) == 1:
                self._opp.copy_(a)
                return
            first = False
            member = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
            for k in slots:
                member |= self.slot == k
            torch.where(member, a, self._opp, out=self._opp)

    def _opening_actions(self, rng=None, pending=None):
        """openings on EMPTY boards: every board that shares a policy has the same distribution, so a
        policy exposing logits() is evaluated on one empty board and sampled once per board; a pool of several
        actor networks is one fused launch over the boards that owe an opening (`pending`, all if None)"""
        n, E = self.board_size, self.num_envs
        tab = self._pool_table()
        if tab is not None and len(tab.policies) > 1:
            tab.actions(self._empty_board, self._all_legal, self.slot, pending, self._opp, self._gen, rng)
            return
        groups = self._groups()
        full = self._all_legal
        for pol, slots in groups.values():
            if hasattr(pol, "logits"):
                # one row, shared by every board (cached per parameter version where the policy offers that)
                z = pol.opening_logits() if hasattr(pol, "opening_logits") else pol.logits(self._empty_board[:1], full[:1])
                if getattr(pol, "deterministic", False):
                    a = z.argmax(1).to(torch.int32).expand(E)
                else:
                    a = sample_masked(z, full, getattr(pol, "generator", None),
                                      pol.bad_rows_counter(self.device) if hasattr(pol, "bad_rows_counter") else None,
                                      out=self._opp if len(groups) == 1 else None, rng=rng if len(groups) == 1 else None)
                    if len(groups) == 1:
                        return
            else:
                a = pol.choose_actions(self._empty_board, full).to(torch.int32)
            if len(groups) == 1:
                self._opp.copy_(a)
                return
            member = torch.zeros(E, dtype=torch.bool, device=self.device)
            for k in slots:
                member |= self.slot == k
            self._opp.copy_(torch.where(member, a, self._opp))

    def _new_positions(self, mask):
        """sample_board: the boards in `mask` were just reset and get a random start position"""
        if not self.sample_board:
            return None
        boards = sample_boards(self.board_size, self.num_envs, self._gen, self.device)
        return self.vec.set_boards(boards, mask)

    def _opponent_round(self, obs, legal, pending, handled, redraw=True, openings_only=False):
        """one round of opponent plies (replies, or openings of freshly reset boards)"""
        if redraw:
            self._setup_opponents(pending == 2)      # fresh boards pick their opponent first (:75, :79-80)
        if openings_only and not self.sample_board:
            self._opening_actions()                  # only empty boards can be pending here
        else:
            self._opponent_actions(obs, legal, pending)
        out = self.vec.step_opponent(self._opp, terminal_obs=self.terminal_obs)
        obs, legal, pending = out[0], out[3], out[5]
        fresh = (self.vec.done != 0) & ~handled
        handled = handled | fresh
        new = self._new_positions(fresh)
        if new is not None:
            obs, legal, pending = new
        return obs, legal, pending, handled

    # -- hexvec_attach_pool ---------------------------------------------------
    def _opening_policies(self):
        """(policies, policy_of_slot int32 [pool + 1] or None) if every pool member has an opening DISTRIBUTION the
        library can tabulate (an actor network's logits on the empty board, or uniform), else None"""
        tab = self._pool_table()
        if tab is not None:
            return tab.policies, (tab.policy_of_slot if len(tab.policies) > 1 else None)
        groups = list(self._groups().values())
        if not all(hasattr(pol, "opening_logits") or isinstance(pol, RandomLegalPolicy) for pol, _ in groups):
            return None
        if len(groups) == 1:
            return [groups[0][0]], None
        key = (id(self.buffer), self.buffer.generation)
        if getattr(self, "_pos_key", None) != key:
            pos = [0] * (len(self.buffer) + 1)
            for g, (_, slots) in enumerate(groups):
                for k in slots:
                    pos[k + 1] = g
            self._pos, self._pos_key = torch.tensor(pos, dtype=torch.int32, device=self.device), key
        return [pol for pol, _ in groups], self._pos

    @torch.no_grad()
    def _refresh_open_cdf(self, policies):
        """cumulative opening distribution per policy (softmax of its logits on the empty board; one-hot at the
        arg-max for a deterministic policy; uniform for the random policy), rebuilt in place when a network changed"""
        C = self.board_size * self.board_size
        vers = [pol.weights_version() if hasattr(pol, "weights_version") else None for pol in policies]
        if self._open_cdf is None or self._open_cdf.shape[0] != len(policies):
            self._open_cdf = torch.empty((len(policies), C), dtype=torch.float32, device=self.device)
            self._open_cdf_versions = None
        if vers == self._open_cdf_versions:
            return self._open_cdf
        for u, pol in enumerate(policies):
            if hasattr(pol, "opening_logits"):
                z = pol.opening_logits().reshape(-1).to(self.device, torch.float64)
                if getattr(pol, "deterministic", False):
                    p = torch.zeros(C, dtype=torch.float64, device=self.device)
                    p[z.argmax()] = 1.0
                    p = torch.where(torch.isnan(z).any(), torch.full_like(p, float("nan")), p)
                else:
                    p = torch.softmax(z, 0)
            else:
                p = torch.full((C,), 1.0 / C, dtype=torch.float64, device=self.device)
            cdf = p.cumsum(0).to(torch.float32)
            # the library's contract: the last entry is exactly 1 for a distribution, anything else marks a row
            # that cannot be sampled (a NaN logit): such boards open at cell 0 and are counted
            ok = torch.isfinite(cdf).all()
            cdf[-1] = torch.where(ok, torch.ones((), device=self.device), torch.full((), float("nan"), device=self.device))
            self._open_cdf[u].copy_(cdf)
        self._open_cdf_versions = vers
        return self._open_cdf

    def _attach_pool(self):
        """make the handle's split step do setup_opponents (and, where the pool allows, the openings) itself;
        returns (attached, openings_inside)"""
        import ctypes as C
        from . import _lib
        want = self.fuse_pool and not self.buffer.evaluating and not self.sample_board
        if not want:
            if self._attached is not None:
                _lib.check(_lib.lib().hexvec_attach_pool(self.vec._h, None, None, C.c_float(0.0), 1, C.c_uint64(0), None,
                                                         None, None, 1, None))
                self._attached = None
            return False, False
        op = self._opening_policies()
        cdf = pos = None
        if op is not None:
            cdf, pos = self._refresh_open_cdf(op[0]), op[1]
        key = (float(self.buffer.prob_best), len(self.buffer), None if cdf is None else cdf.data_ptr(),
               None if pos is None else pos.data_ptr(), self.slot.data_ptr())
        if key != self._attached:
            p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            _lib.check(_lib.lib().hexvec_attach_pool(
                self.vec._h, p(self.slot), p(self._ended), C.c_float(self.buffer.prob_best), len(self.buffer),
                C.c_uint64(self._pick_seed), p(self._pick_state), p(cdf), p(pos),
                1 if cdf is None else int(cdf.shape[0]), p(self._open_bad)))
            self._attached = key
        return True, cdf is not None

    # -- gym-like API ----------------------------------------------------------
    def reset(self):
        """SelfPlayEnv.reset for every board (:69-89): new opponent, then it opens where the agent
        is WHITE."""
        every = torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
        self._draw_candidates()
        self._setup_opponents(every)
        n = self.board_size
        if self.sample_board:            # positions first, then the openings through the split step
            obs, legal, pending = self._new_positions(every)
            self.vec.done.zero_()
            self._opponent_round(obs, legal, pending, every, redraw=False)   # openings, by the opponents chosen above
            return self.vec.obs, self.vec.legal
        # the opponent's view of the empty board is all zeros: its opening is computed up front
        self._opening_actions()
        return self.vec.reset(opp_actions=self._opp)

    def step(self, actions, out=None):
        """actions int [E] (agent's view) -> obs, reward, done, legal, winner.

        Finished boards are reset (with a fresh random position if sample_board); their opponent is
        re-drawn before it opens (if it opens).  No host synchronisation: the number of opponent
        rounds is fixed (replies, then openings if any agent plays WHITE).
        out (optional, training mode with the pool attached to the handle only): the caller's rows for this step's
        results -- any of obs / legal (the NEXT observation and mask) / reward / done / winner, e.g. row t of a rollout
        buffer -- instead of the env's own buffers (HexVecEnv._outs)."""
        attached, openings_inside = self._attach_pool()
        if out and not (attached and (openings_inside or not self._any_white)):
            raise NotImplementedError("step(out=...) needs the training-mode step with the pool attached to the handle "
                                      "(not evaluation mode, sample_board or a pool member without an opening distribution)")
        if self.buffer.evaluating or self.sample_board:
            return self._step_general(actions)
        if attached:
            # two env launches and one policy launch: the slot choice of every board that starts a new episode, the
            # step counter and (openings_inside) the openings of the boards that ended on the reply ride in the tails
            # of hexvec_step_agent / hexvec_step_opponent -- same boards and same draws as the launches below
            # (with `out`: the agent's launch shows the opponents their boards in the env's own buffers -- scratch --
            # and only the step's scalar rows go to the caller; the opponent's launch writes the caller's next rows)
            scal = {k: v for k, v in out.items() if k in ("reward", "done", "winner")} if out else None
            obs, legal, pending = self.vec.step_agent(actions, terminal_obs=self.terminal_obs, out=scal)
            self._opponent_actions(obs, legal, pending, rng=self._rng(3))   # replies (+ openings of boards that just ended)
            res = self.vec.step_opponent(self._opp, terminal_obs=self.terminal_obs, out=out)
            if self._any_white and not openings_inside:     # a pool member without a tabulated opening distribution
                self._opening_actions(rng=self._rng(6), pending=self.vec.pending)
                res = self.vec.step_opponent(self._opp, terminal_obs=self.terminal_obs)
            return res[0], res[1], res[2], res[3], res[4]
        # the same step with every slot update as a launch of its own, the boards that owe no ply are skipped
        # inside the sampler, nothing else is computed on the host side of the ABI
        obs, legal, pending = self.vec.step_agent(actions, terminal_obs=self.terminal_obs)
        # (all draws of the step share one device counter under salts 1..6; the last call advances it)
        # one launch (when = 3): BLACK agents whose game just ended get a new opponent (salt 1); fresh boards pick
        # theirs before it opens (salt 2)
        self._pick(3, salt=1)
        self._opponent_actions(obs, legal, pending, rng=self._rng(3))   # replies (+ openings of boards that just ended)
        out = self.vec.step_opponent(self._opp, terminal_obs=self.terminal_obs)
        if self._any_white:
            # one launch (when = 4): boards that ended on the reply -- WHITE agents' before the opening (salt 4),
            # BLACK agents' for their next episode (salt 5).  The step's counter moves on behind it (every step must
            # advance it, whatever the policies do); the openings then draw under the new value and a salt of their
            # own: (counter, salt) pairs never repeat
            self._pick(4, salt=4, advance=True)
            self._opening_actions(rng=self._rng(6), pending=self.vec.pending)
            self.vec.step_opponent(self._opp, terminal_obs=self.terminal_obs)
        else:
            self._pick(2, salt=5, advance=True)         # ended on the reply, agent BLACK
        return self.vec.obs, self.vec.reward, self.vec.done, self.vec.legal, self.vec.winner

    def _step_general(self, actions):
        """step() in torch ops: evaluation mode (the pool is walked in order: a running count over the boards)
        and sample_board (fresh start positions between the rounds)"""
        self._draw_candidates()
        obs, legal, pending = self.vec.step_agent(actions, terminal_obs=self.terminal_obs)
        ended1 = self.vec.done != 0                    # ended on the agent's ply
        handled = ended1
        new = self._new_positions(ended1)
        if new is not None:
            obs, legal, pending = new
        self._setup_opponents(ended1 & (pending == 0))  # BLACK agents: a new episode, a new opponent
        obs, legal, pending, handled = self._opponent_round(obs, legal, pending, handled)   # replies
        if self._any_white:
            obs, legal, pending, handled = self._opponent_round(obs, legal, pending, handled,
                                                                 openings_only=True)   # openings
        self._setup_opponents((self.vec.done != 0) & ~ended1 & ~self._white)
        return self.vec.obs, self.vec.reward, self.vec.done, self.vec.legal, self.vec.winner

    def _pool_version(self):
        """everything host-side that a captured collection step bakes in: the pool's policy objects and
        their grouping, the evaluation flag, the 80/20 split"""
        b = self.buffer
        return (id(b), b.generation, bool(b.evaluating), float(b.prob_best), self._any_white, bool(self.fuse_pool))

    def capture_collect_step(self, agent_policy, warmup=3):
        """Capture  actions = agent_policy.choose_actions(obs, legal); step(actions)  as ONE hipGraph
        (torch.cuda.CUDAGraph).  step() never synchronises and all shapes are static, so the ~40 small
        launches of a collection step replay with a single host call.  Returns run() -> (obs, reward,
        done, legal, winner, actions); the tensors are static buffers overwritten by every run().
        The policies must be capturable (no host synchronisation, default or registered generators).

        The graph bakes in host-side state: WHICH policy objects are in the pool (and how slots group),
        the evaluation flag and the 80/20 split.  set_eval() / set_opponent_model() / any change of
        `buffer` therefore invalidate it: run() notices (the pool is fingerprinted at capture) and
        captures again before replaying, so the reference's SelfPlayCallback cycle (set_eval ->
        evaluate -> set_opponent_model, SelfplayWrapper.py:113-136) always plays the current opponents;
        `run.captures` counts the captures."""
        holder = {"captures": 0}

        def body():
            if getattr(agent_policy, "takes_row_active", False):    # our policies: draw from the step's counter (salt 7)
                a = agent_policy.choose_actions(self.vec.obs, self.vec.legal, rng=self._rng(7))
            else:
                a = agent_policy.choose_actions(self.vec.obs, self.vec.legal)
            holder["actions"] = a
            self.step(a)

        def capture():
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    body()
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                if hasattr(graph, "register_generator_state"):
                    graph.register_generator_state(self._gen)
                with torch.cuda.graph(graph, stream=side):
                    body()
            torch.cuda.current_stream(self.device).wait_stream(side)
            holder["graph"], holder["version"] = graph, self._pool_version()
            holder["captures"] += 1
            self._graph = graph

        capture()

        def sync_weights(force=False):
            """packed weight tables and cached opening logits are static buffers of the graph: refresh them on the
            host side when a live network changed (one tuple comparison per replay when nothing did)"""
            uniq = holder.get("policies")
            if uniq is None:
                seen, uniq = set(), []
                for pol in [agent_policy] + [self.buffer.policy(k) for k in range(-1, len(self.buffer))]:
                    if id(pol) not in seen and hasattr(pol, "weights_version"):
                        seen.add(id(pol))
                        uniq.append(pol)
                holder["policies"] = uniq
            vers = [pol.weights_version() for pol in uniq]
            if not force and vers == holder.get("weights"):
                return
            tab = self._pool_table()
            if tab is not None:
                tab.sync()
            for pol in uniq:
                if hasattr(pol, "sync_weights"):
                    pol.sync_weights()
            op = self._opening_policies() if self._attached is not None else None
            if op is not None:
                self._refresh_open_cdf(op[0])       # in place: the captured launches read this buffer
            holder["weights"] = vers

        def run():
            if holder["version"] != self._pool_version():
                holder["policies"] = None
                capture()
            sync_weights()
            run.captures = holder["captures"]
            holder["graph"].replay()
            return (self.vec.obs, self.vec.reward, self.vec.done, self.vec.legal, self.vec.winner,
                    holder["actions"])

        run.captures = 1
        return run

    def capture_collect(self, agent_policy, n_steps, warmup=2):
        """n_steps collection steps as ONE hipGraph, every step's rows written straight into a rollout buffer (no copy per
        step, one host call per n_steps): what a PPO collector does between two updates (SB3's collect_rollouts with
        n_steps; the reference's scripts: scripts/experiments/*.py:38-47).  Returns run() -> buf, the static buffers
          obs int8 [n, E, N, N], legal uint8 [n, E, N*N]   what the agent saw / could play at step t
          actions int32 [n, E], reward float32 [n, E], done uint8 [n, E], winner int8 [n, E]   of step t
        (overwritten by every run(); the observation after the last step is the env's own vec.obs / vec.legal, where the
        next run() starts).  Same launches, same draws as n_steps replays of capture_collect_step(): tested equal.
        Training mode with the pool attached to the handle only (step(out=...)); pool changes capture again."""
        n, E, N, C = int(n_steps), self.num_envs, self.board_size, self.board_size ** 2
        dev = self.device
        buf = dict(obs=torch.empty((n, E, N, N), dtype=torch.int8, device=dev),
                   legal=torch.empty((n, E, C), dtype=torch.uint8, device=dev),
                   actions=torch.empty((n, E), dtype=torch.int32, device=dev),
                   reward=torch.empty((n, E), dtype=torch.float32, device=dev),
                   done=torch.empty((n, E), dtype=torch.uint8, device=dev),
                   winner=torch.empty((n, E), dtype=torch.int8, device=dev))
        holder = {"captures": 0}

        def body():
            buf["obs"][0].copy_(self.vec.obs)
            buf["legal"][0].copy_(self.vec.legal)
            for t in range(n):
                if getattr(agent_policy, "takes_row_active", False):
                    agent_policy.choose_actions(buf["obs"][t], buf["legal"][t], out=buf["actions"][t], rng=self._rng(7))
                else:
                    buf["actions"][t].copy_(agent_policy.choose_actions(buf["obs"][t], buf["legal"][t]))
                rows = dict(reward=buf["reward"][t], done=buf["done"][t], winner=buf["winner"][t])
                if t + 1 < n:       # the next observation goes where the next step reads it
                    rows.update(obs=buf["obs"][t + 1], legal=buf["legal"][t + 1])
                self.step(buf["actions"][t], out=rows)

        def capture():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    body()
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    body()
            torch.cuda.current_stream(dev).wait_stream(side)
            holder["graph"], holder["version"] = graph, self._pool_version()
            holder["captures"] += 1
            holder["policies"] = None

        capture()

        def sync_weights():
            uniq = holder.get("policies")
            if uniq is None:
                seen, uniq = set(), []
                for pol in [agent_policy] + [self.buffer.policy(k) for k in range(-1, len(self.buffer))]:
                    if id(pol) not in seen and hasattr(pol, "weights_version"):
                        seen.add(id(pol))
                        uniq.append(pol)
                holder["policies"] = uniq
            vers = [pol.weights_version() for pol in uniq]
            if vers == holder.get("weights"):
                return
            tab = self._pool_table()
            if tab is not None:
                tab.sync()
            for pol in uniq:
                if hasattr(pol, "sync_weights"):
                    pol.sync_weights()
            op = self._opening_policies() if self._attached is not None else None
            if op is not None:
                self._refresh_open_cdf(op[0])
            holder["weights"] = vers

        def run():
            if holder["version"] != self._pool_version():
                capture()
            sync_weights()
            run.captures = holder["captures"]
            holder["graph"].replay()
            return buf

        run.captures = 1
        return run

    def degenerate_rows(self, *more_policies):
        """number of sampling rows so far that had no legal cell or a non-finite logit (sample_masked),
        over the pool's policies (+ any given, e.g. the agent's); one host read.  Always 0 in a healthy
        run: boards handed to a policy always have an empty cell."""
        seen, tot = set(), int(self._open_bad.item())
        for pol in [self.buffer.policy(k) for k in range(-1, len(self.buffer))] + list(more_policies):
            c = getattr(pol, "_bad_rows", None)
            if c is not None and id(c) not in seen:
                seen.add(id(c))
                tot += int(c)
        return tot

    def legal_actions(self):
        return self.vec.legal

    # -- buffer API (SelfplayWrapper.py:106-144) -------------------------------
    def set_eval(self, eval_state):
        self._eval_cursor.zero_()
        self.buffer.set_eval(eval_state)

    def get_scores(self):
        return self.buffer.scores

    def get_opponent_models(self):
        return self.buffer.members

    def get_best_mean_reward(self):
        return self.buffer.best_mean_reward

    def set_opponent_model(self, index, model, score):
        self.buffer.replace(index, as_policy(model, self.device), score)

    def save_best_model(self, path=None):
        self.buffer.save_best(path)

    def episode_stats(self):
        return self.vec.episode_stats()
