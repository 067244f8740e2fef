"""GPU parity: libhexvec (HIP, through the C ABI) vs the golden fixtures generated
from the reference and vs the CPU oracle on seeded inputs.  Bit-exact: this is
integer / byte work, every comparison is array_equal."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import hex_oracle as ho  # noqa: E402  (the checker)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hx():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (torch.cuda.is_available() is False)")
    import hex_gym_env_amd
    return hex_gym_env_amd


def _np(t):
    return t.detach().cpu().numpy()


def _unpack(bits, n):
    return np.unpackbits(bits, axis=-1)[..., :n * n]


# ----------------------------------------------------------------------------
# golden fixtures (reference outputs)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("enc,fname", [("A", "engineA.npz"), ("B", "envB.npz")])
def test_apply_moves_matches_reference_fixtures(hx, enc, fname):
    d = np.load(os.path.join(GOLDEN, fname))
    for n in d["sizes"]:
        n = int(n)
        if n > 14:
            continue
        G = int(d[f"N{n}_ngames"])
        for first in ((0, 1) if enc == "A" else (0,)):
            games = [g for g in range(G) if enc == "B" or int(d[f"N{n}_g{g}_first"]) == first]
            T = max(len(d[f"N{n}_g{g}_actions"]) for g in games)
            acts = np.full((T, len(games)), -1, np.int32)      # -1 = illegal ply: regions unchanged
            for i, g in enumerate(games):
                a = d[f"N{n}_g{g}_actions"]
                acts[:len(a), i] = a
            env = hx.HexVecEnv(n, len(games), encoding=enc, two_sided=True, auto_reset=False,
                               first_player=first)
            ret, done, rb, legal, obs = (_np(x) for x in env.apply_moves(acts))
            _, labels, _ = env.get_state()
            labels = _np(labels)
            for i, g in enumerate(games):
                k = f"N{n}_g{g}_"
                L = len(d[k + "actions"])
                assert np.array_equal(ret[:L, i], d[k + "ret"]), (k, "ret")
                assert np.array_equal(done[:L, i], d[k + "done"]), (k, "done")
                assert np.array_equal(legal[:L, i], _unpack(d[k + "mask"], n)), (k, "mask")
                if enc == "A":
                    assert np.array_equal(obs[:L, i], d[k + "board"]), (k, "board")
                else:
                    assert np.array_equal(obs[:L, i], d[k + "obs"]), (k, "obs")
                    assert np.array_equal(rb[:L, i], d[k + "r_black"]), (k, "r_black")
                assert np.array_equal(labels[i].reshape(2, n + 2, n + 2), d[k + "regions"]), (k, "regions")
            env.close()


def test_selfplay_step_matches_reference_fixtures(hx):
    d = np.load(os.path.join(GOLDEN, "selfplay.npz"))
    for n in d["sizes"]:
        n = int(n)
        for agent in (0, 1):
            eps = [e for e in range(int(d[f"N{n}_neps"])) if int(d[f"N{n}_e{e}_agent"]) == agent]
            E = len(eps)
            env = hx.HexVecEnv(n, E, encoding="B", agent_player_num=agent, auto_reset=False)
            ro = np.array([max(int(d[f"N{n}_e{e}_reset_opp_action"]), 0) for e in eps], np.int32)
            obs, legal = env.reset(opp_actions=ro)
            obs, legal = _np(obs).reshape(E, -1), _np(legal)
            for i, e in enumerate(eps):
                k = f"N{n}_e{e}_"
                assert np.array_equal(obs[i], d[k + "reset_obs"]), k
                assert np.array_equal(legal[i], _unpack(d[k + "reset_mask"], n)), k
            T = max(len(d[f"N{n}_e{e}_actions"]) for e in eps)
            for t in range(T):
                a = np.zeros(E, np.int32)
                oa = np.zeros(E, np.int32)
                for i, e in enumerate(eps):
                    k = f"N{n}_e{e}_"
                    if t < len(d[k + "actions"]):
                        a[i] = d[k + "actions"][t]
                        oa[i] = max(int(d[k + "opp_actions"][t]), 0)
                obs, rew, done, legal, win = env.step(a, oa)
                obs, rew, done, legal = _np(obs).reshape(E, -1), _np(rew), _np(done), _np(legal)
                for i, e in enumerate(eps):
                    k = f"N{n}_e{e}_"
                    if t < len(d[k + "actions"]):
                        assert np.array_equal(obs[i], d[k + "obs"][t]), (k, t)
                        assert rew[i] == d[k + "reward"][t], (k, t)
                        assert done[i] == d[k + "done"][t], (k, t)
                        assert np.array_equal(legal[i], _unpack(d[k + "mask"], n)[t]), (k, t)
            env.close()


def test_hexv0_step_matches_reference_fixtures(hx):
    d = np.load(os.path.join(GOLDEN, "envA.npz"))
    for n in d["sizes"]:
        n = int(n)
        for first in (0, 1):
            eps = [e for e in range(int(d[f"N{n}_neps"])) if int(d[f"N{n}_e{e}_first"]) == first]
            E = len(eps)
            env = hx.HexVecEnv(n, E, encoding="A", auto_reset=False, first_player=first)
            ro = np.array([max(int(d[f"N{n}_e{e}_reset_opp_action"]), 0) for e in eps], np.int32)
            obs, legal = env.reset(opp_actions=ro)
            obs, legal = _np(obs).reshape(E, -1), _np(legal)
            for i, e in enumerate(eps):
                k = f"N{n}_e{e}_"
                assert np.array_equal(obs[i], d[k + "reset_obs"]), k
                assert np.array_equal(legal[i], _unpack(d[k + "reset_mask"], n)), k
            T = max(len(d[f"N{n}_e{e}_actions"]) for e in eps)
            for t in range(T):
                a = np.zeros(E, np.int32)
                oa = np.zeros(E, np.int32)
                for i, e in enumerate(eps):
                    k = f"N{n}_e{e}_"
                    if t < len(d[k + "actions"]):
                        a[i] = d[k + "actions"][t]
                        oa[i] = max(int(d[k + "opp_actions"][t]), 0)
                obs, rew, done, legal, win = env.step(a, oa)
                obs, rew, done, legal, win = (_np(obs).reshape(E, -1), _np(rew), _np(done),
                                              _np(legal), _np(win))
                for i, e in enumerate(eps):
                    k = f"N{n}_e{e}_"
                    if t < len(d[k + "actions"]):
                        assert np.array_equal(obs[i], d[k + "obs"][t]), (k, t)
                        assert rew[i] == d[k + "reward"][t], (k, t)
                        assert done[i] == d[k + "done"][t], (k, t)
                        assert win[i] == d[k + "winner"][t], (k, t)
                        assert np.array_equal(legal[i], _unpack(d[k + "mask"], n)[t]), (k, t)
            env.close()


def _replay_episodes_auto_reset(hx, d, n, eps, enc, lay, **kw):
    """Fixture episodes through an AUTO-RESET handle (the throughput kernels): every ply of every
    episode up to its own end is compared (obs, mask, reward, done, winner); the final ply's
    observation is the terminal_obs row, because the returned obs already shows the next episode."""
    E = len(eps)
    env = hx.HexVecEnv(n, E, encoding=enc, rollout_layout=lay == "lpb", **kw)
    ro = np.array([max(int(d[f"N{n}_e{e}_reset_opp_action"]), 0) for e in eps], np.int32)
    obs, legal = env.reset(opp_actions=ro)
    if n >= 4:
        assert env.last_kernel().startswith("k_lpb" if lay == "lpb" else "k_qpb")
    obs, legal = _np(obs).reshape(E, -1), _np(legal)
    for i, e in enumerate(eps):
        k = f"N{n}_e{e}_"
        assert np.array_equal(obs[i], d[k + "reset_obs"]), k
        assert np.array_equal(legal[i], _unpack(d[k + "reset_mask"], n)), k
    T = max(len(d[f"N{n}_e{e}_actions"]) for e in eps)
    over = np.zeros(E, bool)
    tobs = torch.full((E, n, n), 99, dtype=torch.int8, device="cuda")
    empty = 0 if enc == "B" else 2
    checked = 0
    for t in range(T):
        a = np.zeros(E, np.int32)
        oa = np.zeros(E, np.int32)
        for i, e in enumerate(eps):
            k = f"N{n}_e{e}_"
            if t < len(d[k + "actions"]):
                a[i] = d[k + "actions"][t]
                oa[i] = max(int(d[k + "opp_actions"][t]), 0)
        obs, rew, done, legal, win = env.step(a, oa, terminal_obs=tobs)
        obs, rew, done, legal, win = _np(obs).reshape(E, -1), _np(rew), _np(done), _np(legal), _np(win)
        tb = _np(tobs).reshape(E, -1)
        for i, e in enumerate(eps):
            k = f"N{n}_e{e}_"
            if over[i] or t >= len(d[k + "actions"]):
                continue
            assert rew[i] == d[k + "reward"][t], (k, t)
            assert done[i] == d[k + "done"][t], (k, t)
            if k + "winner" in d:
                assert win[i] == d[k + "winner"][t], (k, t)
            m = _unpack(d[k + "mask"], n)[t]
            if done[i]:
                assert np.array_equal(tb[i], d[k + "obs"][t]), (k, t, "terminal obs")
                assert np.array_equal((tb[i] == empty).astype(np.uint8), m), (k, t)
                over[i] = True
            else:
                assert np.array_equal(obs[i], d[k + "obs"][t]), (k, t)
                assert np.array_equal(legal[i], m), (k, t)
            checked += 1
    env.close()
    return checked


LAYOUTS = pytest.mark.parametrize("lay", [False, "lpb"], ids=["bitboard", "lane-per-board"])


@LAYOUTS
def test_selfplay_fixtures_on_auto_reset_handles(hx, lay):
    """SelfPlayEnv episodes of the reference (SelfplayWrapper.py:174-199) on the throughput kernels"""
    d = np.load(os.path.join(GOLDEN, "selfplay.npz"))
    tot = 0
    for n in (int(x) for x in d["sizes"]):
        for agent in (0, 1):
            eps = [e for e in range(int(d[f"N{n}_neps"])) if int(d[f"N{n}_e{e}_agent"]) == agent]
            tot += _replay_episodes_auto_reset(hx, d, n, eps, "B", lay, agent_player_num=agent)
    assert tot > 200


@LAYOUTS
def test_hexv0_fixtures_on_auto_reset_handles(hx, lay):
    """hex-v0 HexEnv episodes of the reference (HexGame.py:244-295) on the throughput kernels"""
    d = np.load(os.path.join(GOLDEN, "envA.npz"))
    tot = 0
    for n in (int(x) for x in d["sizes"]):
        for first in (0, 1):
            eps = [e for e in range(int(d[f"N{n}_neps"])) if int(d[f"N{n}_e{e}_first"]) == first]
            tot += _replay_episodes_auto_reset(hx, d, n, eps, "A", lay, first_player=first)
    assert tot > 100


# ----------------------------------------------------------------------------
# oracle on seeded inputs
# ----------------------------------------------------------------------------
def _flags(enc, two_sided, agent, auto_reset, first):
    f = 0
    if enc == "B":
        f |= ho.F_ENC_B
        if agent is None:
            f |= ho.F_AGENT_RANDOM
        elif agent == 1:
            f |= ho.F_AGENT_WHITE
    elif first == 1:
        f |= ho.F_FIRST_WHITE
    if two_sided:
        f |= ho.F_TWO_SIDED
    if not auto_reset:
        f |= ho.F_NO_AUTO_RESET
    return f


# exact: False = bitboard kernels (default), "lpb" = lane-per-board union-find (rollout layout),
# True = wave-per-board literal engine
FAMILIES = pytest.mark.parametrize("exact", [False, "lpb", True],
                                   ids=["bitboard", "lane-per-board", "wave-per-board"])


def _pair(hx, n, E, enc, two_sided=False, agent=None, auto_reset=True, first=0, seed=7, off=0,
          exact=False):
    lpb, exact = exact == "lpb", exact is True
    g = hx.HexVecEnv(n, E, encoding=enc, two_sided=two_sided, agent_player_num=agent,
                     auto_reset=auto_reset, first_player=first, seed=seed, env_offset=off,
                     exact_labels=exact, rollout_layout=lpb)
    assert g.exact == (exact or not auto_reset or n < 4)
    assert g.last_kernel().startswith("k_hex" if g.exact else ("k_lpb" if lpb else "k_qpb"))
    o = ho.OracleVec(n, E, seed=seed, flags=_flags(enc, two_sided, agent, auto_reset, first),
                     env_offset=off)
    return g, o


def _same_state(g, o):
    gb, gl, gc = g.get_state()
    ob, ol, oc = o.get_state()
    assert np.array_equal(_np(gb), ob), "board"
    if g.exact:
        assert np.array_equal(_np(gl), ol), "labels"
    else:
        assert gl is None
    assert np.array_equal(_np(gc), oc), "cur"
    assert np.array_equal(_np(g.episode_stats()), o.episode_stats()), "episode stats"


@pytest.mark.parametrize("n", [2, 3, 5, 7, 8, 9, 11, 12, 13, 14])
@pytest.mark.parametrize("enc", ["A", "B"])
def test_apply_moves_random_cells_vs_oracle(hx, n, enc):
    """uniformly random cells: many illegal plies, plies after the end, full boards."""
    E, T = 96, 3 * n * n + 5
    rng = np.random.default_rng(100 * n + (enc == "B"))
    acts = rng.integers(0, n * n, size=(T, E)).astype(np.int32)
    g, o = _pair(hx, n, E, enc, two_sided=True, auto_reset=False, first=n % 2 if enc == "A" else 0)
    gr = [_np(x) for x in g.apply_moves(acts)]
    orr = o.apply_moves(acts)
    for name, a, b in zip(("ret", "done", "r_black", "legal", "obs"), gr, orr):
        assert np.array_equal(a, b), name
    _same_state(g, o)


@pytest.mark.parametrize("n", [3, 5, 7, 11, 13])
@pytest.mark.parametrize("enc,agent,first", [("B", None, 0), ("B", 0, 0), ("B", 1, 0),
                                             ("A", None, 0), ("A", None, 1)])
@FAMILIES
def test_selfplay_rng_steps_vs_oracle(hx, n, enc, agent, first, exact):
    """W1: RNG agent + RNG opponent + auto reset, many episodes."""
    E, K = 64, 6 * n
    g, o = _pair(hx, n, E, enc, agent=agent, first=first, seed=1234 + n, exact=exact)
    if enc == "B":
        assert np.array_equal(_np(g.agent_colours()), o.agent_colours())
    tob = torch.zeros((E, n, n), dtype=torch.int8, device="cuda")
    for k in range(K):
        obs, rew, done, legal, win = g.step(terminal_obs=tob)
        oo, ol, orw, od, ow, otob = o.step(want_terminal=True)
        assert np.array_equal(_np(obs).reshape(E, -1), oo), k
        assert np.array_equal(_np(legal), ol), k
        assert np.array_equal(_np(rew), orw), k
        assert np.array_equal(_np(done), od), k
        assert np.array_equal(_np(win), ow), k
        m = od.astype(bool)
        assert np.array_equal(_np(tob).reshape(E, -1)[m], otob[m]), k
    assert o.episode_stats()[2] >= E // 2    # many finished episodes (incl. auto resets)
    _same_state(g, o)


@FAMILIES
@pytest.mark.parametrize("n", [4, 7, 11])
@pytest.mark.parametrize("enc", ["A", "B"])
def test_selfplay_external_actions_vs_oracle(hx, n, enc, exact):
    """external agent actions (random cells, some illegal) + RNG opponent."""
    E, K = 64, 5 * n
    rng = np.random.default_rng(n)
    g, o = _pair(hx, n, E, enc, agent=None, seed=99, exact=exact)
    for k in range(K):
        a = rng.integers(0, n * n, size=E).astype(np.int32)
        obs, rew, done, legal, win = g.step(a)
        oo, ol, orw, od, ow = o.step(a)
        assert np.array_equal(_np(obs).reshape(E, -1), oo), k
        assert np.array_equal(_np(legal), ol), k
        assert np.array_equal(_np(rew), orw), k
        assert np.array_equal(_np(done), od), k
        assert np.array_equal(_np(win), ow), k
    _same_state(g, o)


@FAMILIES
@pytest.mark.parametrize("n", [3, 6, 11, 14])
@pytest.mark.parametrize("enc", ["A", "B"])
def test_two_sided_steps_and_masked_reset_vs_oracle(hx, n, enc, exact):
    E, K = 64, 4 * n
    rng = np.random.default_rng(n + 50)
    g, o = _pair(hx, n, E, enc, two_sided=True, seed=5, first=1 if enc == "A" else 0, exact=exact)
    for k in range(K):
        a = rng.integers(0, n * n, size=E).astype(np.int32) if k % 3 else None
        obs, rew, done, legal, win = g.step(a)
        oo, ol, orw, od, ow = o.step(a)
        assert np.array_equal(_np(obs).reshape(E, -1), oo), k
        assert np.array_equal(_np(legal), ol), k
        assert np.array_equal(_np(rew), orw), k
        assert np.array_equal(_np(done), od), k
        assert np.array_equal(_np(win), ow), k
        if k % 7 == 3:
            m = (rng.random(E) < 0.3).astype(np.uint8)
            gobs, glegal = g.reset(env_mask=m)
            oobs, olegal = o.reset(env_mask=m)
            assert np.array_equal(_np(gobs).reshape(E, -1), oobs)
            assert np.array_equal(_np(glegal), olegal)
    _same_state(g, o)


@FAMILIES
@pytest.mark.parametrize("n", [2, 5, 8, 11, 13, 14])
@pytest.mark.parametrize("enc", ["A", "B"])
def test_rollout_random_vs_oracle(hx, n, enc, exact):
    """W2: device-resident rollout, every ply recorded, vs the oracle's loop."""
    E, T = 80, 2 * n * n + 7
    g, o = _pair(hx, n, E, enc, two_sided=True, seed=2024, exact=exact)
    out = g.rollout_random(T)
    oobs, olegal, oact, orew, odone = o.rollout(T)
    assert np.array_equal(_np(out["obs"]), oobs)
    assert np.array_equal(_np(out["legal"]), olegal)
    assert np.array_equal(_np(out["actions"]), oact)
    assert np.array_equal(_np(out["reward"]), orew)
    assert np.array_equal(_np(out["done"]), odone)
    _same_state(g, o)
    # a second launch continues from the stored state
    out = g.rollout_random(9)
    o2 = o.rollout(9)
    assert np.array_equal(_np(out["obs"]), o2[0])
    assert np.array_equal(_np(out["done"]), o2[4])
    _same_state(g, o)


@FAMILIES
@pytest.mark.parametrize("n", [5, 11])
@pytest.mark.parametrize("enc", ["A", "B"])
def test_rollout_selfplay_vs_oracle(hx, n, enc, exact):
    E, T = 64, 4 * n
    g, o = _pair(hx, n, E, enc, agent=None, seed=31, exact=exact)
    out = g.rollout_selfplay(T)
    for t in range(T):
        oo, ol, orw, od, ow = o.step()
        assert np.array_equal(_np(out["obs"][t]), oo), t
        assert np.array_equal(_np(out["legal"][t]), ol), t
        assert np.array_equal(_np(out["reward"][t]), orw), t
        assert np.array_equal(_np(out["done"][t]), od), t
    _same_state(g, o)


# the kernel bench.py times, at the geometry it is timed at (many workgroups, XCD block remap, a ragged tail block
# beside full ones, the helper wavefront's scalar path, in-kernel resets + openings, the LDS-DMA state reload across
# launches): SelfplayWrapper.py:69-89, :174-199 through k_lpb<rollout1>, every recorded row against the oracle
@pytest.mark.parametrize("n,E,enc,first,plan", [
    (11, 4096, "B", 0, [20] * 10 + [200]),          # 64 workgroups = 8 per XCD; the driver's 20-step launches, then 200
    (11, 4096 + 37, "B", 0, [20] * 10 + [200]),     # + a tail block (helper off) beside 64 full ones, odd block count
    (13, 1024, "B", 0, [20] * 10 + [200, 100]),     # one image per board at the LDS limit
    (5, 64 * 9 + 17, "B", 0, [7, 20, 33, 20, 20, 50]),
    (7, 64 * 16, "B", 0, [20] * 5 + [100]),
    (5, 64 * 9 + 17, "A", 1, [7, 20, 33, 20, 20, 50]),  # hex-v0, the opponent opens every episode
    (11, 64 * 8, "A", 0, [20] * 5 + [150]),
], ids=lambda v: None if isinstance(v, list) else str(v))
def test_rollout1_at_bench_geometry_every_row_vs_oracle(hx, n, E, enc, first, plan):
    g, o = _pair(hx, n, E, enc, agent=None, first=first, seed=20261005, exact="lpb")
    o.set_threads(min(os.cpu_count() or 1, 64))
    finished = np.zeros(E, np.int64)
    for launch, T in enumerate(plan):
        out = g.rollout_selfplay(T)
        assert g.last_kernel() == "k_lpb<rollout1>"
        oobs, olegal, oact, orew, odone = o.rollout_selfplay(T)
        for k, ref in (("obs", oobs), ("legal", olegal), ("actions", oact), ("reward", orew), ("done", odone)):
            got = _np(out[k])
            if not np.array_equal(got, ref):
                t, e = np.argwhere((got != ref).reshape(T, E, -1).any(-1))[0]
                raise AssertionError(f"{k}: launch {launch} (T={T}) first differs at step {t}, env {e} (block {e // 64})")
        finished += odone.sum(0, dtype=np.int64)
        assert np.array_equal(_np(g.episode_stats()), o.episode_stats()), launch
    assert finished.min() >= 3, "every env must have run through several in-kernel resets"
    _same_state(g, o)


# ----------------------------------------------------------------------------
# full-size properties (BASELINE.json configs[1]: 11x11, 65 536 envs)
# ----------------------------------------------------------------------------
def test_full_size_properties(hx):
    n, E, T = 11, 65536, 32
    g = hx.HexVecEnv(n, E, encoding="B", two_sided=True, seed=3)
    out = g.rollout_random(T)
    obs, legal, act, rew, done = (out[k] for k in ("obs", "legal", "actions", "reward", "done"))
    # mask == empties of the obs it was recorded with
    assert torch.equal(legal, (obs == 0).to(torch.uint8))
    # the recorded action is always a legal cell
    assert bool(torch.gather(legal, 2, act.long().unsqueeze(-1)).all())
    # a mover's reward is +1 exactly on the plies that end a game (no draws in Hex)
    assert torch.equal(rew, done.float())
    # stones on the board before ply t == plies since the last reset
    stones = (obs != 0).sum(-1)
    since = torch.zeros(E, dtype=torch.long, device="cuda")
    for t in range(T):
        assert torch.equal(stones[t], since)
        since = torch.where(done[t].bool(), torch.zeros_like(since), since + 1)
    # equal stone counts for both colours up to one
    diff = (obs == -1).sum(-1) - (obs == 1).sum(-1)
    assert int(diff.abs().max()) <= 1
    st = _np(g.episode_stats())
    assert st[2] == float(done.sum())
    # sharding invariance: the upper half as its own shard (multi-GPU layout)
    h = hx.HexVecEnv(n, E // 2, encoding="B", two_sided=True, seed=3, env_offset=E // 2)
    out2 = h.rollout_random(T)
    assert torch.equal(out2["obs"], obs[:, E // 2:])
    assert torch.equal(out2["actions"], act[:, E // 2:])
    # mean random game length (SURVEY: 107.4 plies at 11x11)
    g2 = hx.HexVecEnv(n, 4096, encoding="B", two_sided=True, seed=11)
    g2.rollout_random(600, out={})
    s = _np(g2.episode_stats())
    assert 100.0 < s[1] / s[2] < 115.0


def test_selfplay_full_size_step_properties(hx):
    n, E = 11, 65536
    g = hx.HexVecEnv(n, E, encoding="B", seed=9)
    col = g.agent_colours()
    assert 0.45 < float(col.float().mean()) < 0.55
    tot_done = 0
    for k in range(80):
        obs, rew, done, legal, win = g.step()
        assert torch.equal(legal, (obs.reshape(E, -1) == 0).to(torch.uint8))
        assert bool(((rew != 0) <= done.bool()).all())
        tot_done += int(done.sum())
    st = _np(g.episode_stats())
    assert st[2] == tot_done and tot_done > E
    assert abs(st[0]) < 0.1 * st[2]        # random vs random: about even


def test_errors_are_loud(hx):
    with pytest.raises(ValueError):
        hx.HexVecEnv(20, 4)
    for kw in (dict(auto_reset=False), dict(exact_labels=True), dict(rollout_layout=True)):
        with pytest.raises(ValueError):
            hx.HexVecEnv(15, 4, **kw)            # labels / union-find entries are one byte: 2..14
    with pytest.raises(NotImplementedError):
        hx.HexVecEnv(5, 4, encoding="A", agent_player_num=1)
    e = hx.HexVecEnv(5, 4, encoding="B")
    with pytest.raises(RuntimeError):
        e.rollout_random(4)              # needs a two-sided handle
    with pytest.raises(ValueError):
        e.step(np.zeros(3, np.int32))


@pytest.mark.parametrize("n,E", [(5, 70), (11, 65), (7, 1), (4, 129), (13, 9), (8, 33)])
@pytest.mark.parametrize("enc", ["A", "B"])
@LAYOUTS
def test_lane_per_board_odd_env_counts(hx, n, E, enc, lay):
    """partial last wavefront and unaligned output rows"""
    T = 3 * n
    g, o = _pair(hx, n, E, enc, two_sided=True, seed=77, exact=lay)
    out = g.rollout_random(T)
    oobs, olegal, oact, orew, odone = o.rollout(T)
    assert np.array_equal(_np(out["obs"]), oobs)
    assert np.array_equal(_np(out["legal"]), olegal)
    assert np.array_equal(_np(out["actions"]), oact)
    assert np.array_equal(_np(out["done"]), odone)
    g2, o2 = _pair(hx, n, E, enc, agent=None, seed=78, exact=lay)
    for k in range(2 * n):
        obs, rew, done, legal, win = g2.step()
        oo, ol, orw, od, ow = o2.step()
        assert np.array_equal(_np(obs).reshape(E, -1), oo), k
        assert np.array_equal(_np(legal), ol), k
        assert np.array_equal(_np(rew), orw), k
        assert np.array_equal(_np(done), od), k
    _same_state(g2, o2)


@pytest.mark.parametrize("n", [15, 16, 17, 19])
def test_large_boards_on_the_bitboard_layout_vs_oracle(hx, n):
    """15x15..19x19 (8 or 16 lanes per board): self-play steps with mixed RNG / external actions, both
    encodings, then a two-sided rollout -- all rows against the oracle"""
    E, K = 96, 3 * n * n // 2
    rng = np.random.default_rng(n)
    for enc, first in (("B", 0), ("A", n % 2)):
        g, o = _pair(hx, n, E, enc, agent=None, first=first, seed=77 + n)
        for k in range(K):
            a = rng.integers(0, n * n, size=E).astype(np.int32) if k % 6 == 5 else None
            obs, rew, done, legal, win = g.step(a)
            oo, ol, orw, od, ow = o.step(a)
            assert np.array_equal(_np(obs).reshape(E, -1), oo), (enc, k)
            assert np.array_equal(_np(legal), ol), (enc, k)
            assert np.array_equal(_np(rew), orw) and np.array_equal(_np(done), od), (enc, k)
            assert np.array_equal(_np(win), ow), (enc, k)
        assert o.episode_stats()[2] > E // 2
        _same_state(g, o)
    g, o = _pair(hx, n, 40, "B", two_sided=True, seed=5 + n)
    T = n * n + 9
    out = g.rollout_random(T)
    oobs, olegal, oact, orew, odone = o.rollout(T)
    for k, ref in (("obs", oobs), ("legal", olegal), ("actions", oact), ("reward", orew), ("done", odone)):
        assert np.array_equal(_np(out[k]), ref), k
    _same_state(g, o)


@pytest.mark.parametrize("n", list(range(4, 15)))
@LAYOUTS
def test_soak_all_sizes_lane_per_board_vs_oracle(hx, n, lay):
    """long differential run per board size: hundreds of episodes per env, both encodings, RNG and
    external (partly illegal) actions mixed, masked resets in between"""
    E, K = 256, 300
    rng = np.random.default_rng(1000 + n)
    for enc, agent, first in (("B", None, 0), ("A", None, n % 2)):
        g, o = _pair(hx, n, E, enc, agent=agent, first=first, seed=4242 + n, exact=lay)
        for k in range(K):
            if k % 5 == 4:
                a = rng.integers(0, n * n, size=E).astype(np.int32)
            else:
                a = None
            obs, rew, done, legal, win = g.step(a)
            oo, ol, orw, od, ow = o.step(a)
            assert np.array_equal(_np(rew), orw) and np.array_equal(_np(done), od), (enc, k)
            assert np.array_equal(_np(win), ow), (enc, k)
            assert np.array_equal(_np(obs).reshape(E, -1), oo), (enc, k)
            assert np.array_equal(_np(legal), ol), (enc, k)
            if k % 97 == 50:
                m = (rng.random(E) < 0.2).astype(np.uint8)
                gobs, glegal = g.reset(env_mask=m)
                oobs, olegal = o.reset(env_mask=m)
                assert np.array_equal(_np(gobs).reshape(E, -1), oobs), (enc, k)
                assert np.array_equal(_np(glegal), olegal), (enc, k)
        assert o.episode_stats()[2] > 3 * E
        _same_state(g, o)


@LAYOUTS
def test_soak_two_sided_rollout_long(hx, lay):
    """W2 over many launches: state carried across launches, ~10^2 episodes per env"""
    n, E = 9, 512
    g, o = _pair(hx, n, E, "B", two_sided=True, seed=606, exact=lay)
    for launch in range(12):
        T = 37 + launch
        out = g.rollout_random(T, out={k: torch.empty((T, E) + ((n * n,) if k in ("obs", "legal") else ()),
                                                      dtype=dt, device="cuda")
                                       for k, dt in (("obs", torch.int8), ("actions", torch.int32),
                                                     ("done", torch.uint8))})
        oobs, olegal, oact, orew, odone = o.rollout(T)
        assert np.array_equal(_np(out["actions"]), oact), launch
        assert np.array_equal(_np(out["done"]), odone), launch
        assert np.array_equal(_np(out["obs"][-1]), oobs[-1]), launch
    _same_state(g, o)


@pytest.mark.parametrize("enc", ["A", "B"])
def test_out_of_range_actions_are_illegal_plies_on_both_families(hx, enc):
    """the reference would raise (or wrap) on an out-of-board index; the ABI defines it as an illegal
    ply: done, reward 0 (env B / self-play) or -100 (hex-v0), winner code 3 -- same on both families"""
    n, E = 6, 128
    bad = np.array([-1, -7, n * n, n * n + 1, 10 ** 6, -2 ** 31, 2 ** 31 - 1, 255], np.int32)
    rng = np.random.default_rng(3)
    fast = hx.HexVecEnv(n, E, encoding=enc, seed=5)
    exact = hx.HexVecEnv(n, E, encoding=enc, seed=5, exact_labels=True)
    for k in range(12):
        a = rng.integers(0, n * n, size=E).astype(np.int32)
        idx = rng.choice(E, size=16, replace=False)
        a[idx] = rng.choice(bad, size=16)
        fo = [x.clone() for x in fast.step(a)]
        eo = exact.step(a)
        for x, y in zip(fo, eo):
            assert torch.equal(x, y), k
        rew, done, win = _np(fo[1]), _np(fo[2]), _np(fo[4])
        assert (done[idx] == 1).all() and (win[idx] == 3).all()
        assert (rew[idx] == (-100 if enc == "A" else 0)).all()
    two_f = hx.HexVecEnv(n, E, encoding=enc, two_sided=True, seed=6)
    two_e = hx.HexVecEnv(n, E, encoding=enc, two_sided=True, seed=6, exact_labels=True)
    a = np.full(E, -3, np.int32)
    for x, y in zip([t.clone() for t in two_f.step(a)], two_e.step(a)):
        assert torch.equal(x, y)


def test_config1_thousand_reference_games_5x5(hx):
    """BASELINE config 1 on the GPU: the 1 000 reference games replayed as 1 000 envs"""
    d = np.load(os.path.join(GOLDEN, "config1_5x5.npz"))
    n, acts, rets = int(d["n"]), d["actions"].astype(np.int32), d["ret"]
    G, T = acts.shape
    env = hx.HexVecEnv(n, G, encoding="A", two_sided=True, auto_reset=False)
    ret, done, _, legal, obs = (_np(x) for x in env.apply_moves(np.ascontiguousarray(acts.T)))
    played = acts.T >= 0
    assert np.array_equal(ret[played], rets.T[played])
    last = played.sum(0) - 1
    assert (done[last, np.arange(G)] == 1).all() and (done[0] == 0).all()
    # and through the single-ply step of both throughput layouts (auto reset on the winning ply): every
    # ply of every game to that game's own end -- winner code, done, observation (the terminal one for
    # the last ply), legal mask -- against the literal engine's rows, which the fixture pins
    for lay in (False, True):
        fast = hx.HexVecEnv(n, G, encoding="A", two_sided=True, rollout_layout=lay)
        tobs = torch.full((G, n, n), 99, dtype=torch.int8, device="cuda")
        alive = np.ones(G, bool)
        for t in range(T):
            a = np.where(acts[:, t] >= 0, acts[:, t], 0).astype(np.int32)
            o, r, dn, lg, w = fast.step(a, terminal_obs=tobs)
            assert fast.last_kernel().startswith("k_lpb" if lay else "k_qpb")
            o, dn, lg, w, tb = _np(o).reshape(G, -1), _np(dn), _np(lg), _np(w), _np(tobs).reshape(G, -1)
            cur = alive & (acts[:, t] >= 0)
            assert np.array_equal(w[cur], rets[cur, t]), t
            assert np.array_equal(dn[cur], done[t, cur]), t
            fin = cur & (dn == 1)
            go = cur & (dn == 0)
            assert np.array_equal(o[go], obs[t, go]) and np.array_equal(lg[go], legal[t, go]), t
            assert np.array_equal(tb[fin], obs[t, fin]), t
            alive &= ~fin
        assert not alive.any()


@LAYOUTS
@pytest.mark.parametrize("two", [False, True], ids=["selfplay", "two-sided"])
@pytest.mark.parametrize("n,E", [(11, 524288), (13, 131072)], ids=["config4", "config5"])
def test_full_job_on_one_gpu_and_shard_identity(hx, lay, two, n, E):
    """BASELINE config 4 (11x11, 524 288 envs = 8 x 65 536) and config 5's env side (13x13, 131 072 envs = 8 x 16 384):
    the whole job on ONE GPU -- size-independent properties on every row -- and sharding: rank 7's shard
    (env_offset = 7E/8) reproduces the last eighth of the monolithic run bit for bit, and its first envs match the
    oracle started at the same global ids.  (Config 5's policy in the loop: test_baseline_config_sizes_selfplay_with_mlp_policy.)"""
    T = 8
    kw = dict(encoding="B", two_sided=two, seed=2, rollout_layout=lay == "lpb")
    env = hx.HexVecEnv(n, E, **kw)
    fn = env.rollout_random if two else env.rollout_selfplay
    o1 = fn(T)
    assert env.last_kernel().startswith("k_lpb" if lay == "lpb" else "k_qpb")
    obs, legal, done, act = o1["obs"], o1["legal"], o1["done"], o1["actions"]
    assert torch.equal(legal, (obs == 0).to(torch.uint8))
    assert int(((act < 0) | (act >= n * n)).sum()) == 0
    if two:     # the recorded action was legal in the recorded position
        assert bool(torch.gather(legal, 2, act.long().unsqueeze(-1)).all())
    stones = (obs != 0).sum(-1)
    assert int(stones.max()) <= n * n and int(done.sum()) == 0          # no 11x11 / 13x13 game ends within 8 steps
    st = _np(env.episode_stats())
    assert st[2] == 0
    sh = hx.HexVecEnv(n, E // 8, env_offset=7 * E // 8, **kw)
    o2 = (sh.rollout_random if two else sh.rollout_selfplay)(T)
    for k in ("obs", "legal", "actions", "reward", "done"):
        assert torch.equal(o1[k][:, 7 * E // 8:], o2[k]), k
    Eo = 512
    o = ho.OracleVec(n, Eo, seed=2, flags=_flags("B", two, None, True, 0), env_offset=7 * E // 8)
    if two:
        oobs, olegal, oact, orew, odone = o.rollout(T)
        assert np.array_equal(_np(o2["obs"][:, :Eo]), oobs) and np.array_equal(_np(o2["actions"][:, :Eo]), oact)
    else:
        for t in range(T):
            oo, ol, orw, od, ow = o.step()
            assert np.array_equal(_np(o2["obs"][t, :Eo]), oo), t
            assert np.array_equal(_np(o2["reward"][t, :Eo]), orw), t


@FAMILIES
def test_episode_sums_are_64_bit(hx, exact):
    """hexvec_episode_stats reports sums since create: sum_len grows by ~3e5 per second and env in a rollout and passed
    2^31 within two hours when the per-env sums were 32-bit.  A checkpoint whose per-env sums sit just below 2^31 /
    2^32 is imported, more episodes are played, and the totals must come out exact (two handles, same seed: the
    poked one must read the untouched one's totals + what was poked in).  Blob layout: include/hexvec.h
    (hexvec_export_state) keeps it opaque; the offsets below are this library version's (csrc/hexvec_lpb.h
    FastState, csrc/hexvec_qpb.h k_qpb `stats`, csrc/hexvec.hip export order)."""
    n, E, T = 5, 200, 60
    lpb = exact == "lpb"
    kw = dict(encoding="B", seed=77, exact_labels=exact is True, rollout_layout=lpb)
    a, b = hx.HexVecEnv(n, E, **kw), hx.HexVecEnv(n, E, **kw)
    for _ in range(10):     # some history first
        a.step()
        b.step()
    sd = b.state_dict()
    blob = _np(sd["blob"]).copy()
    add = np.zeros((E, 4), np.int64)       # what the poke adds to (sum_return, sum_len, n_episodes, n_wins) per env
    if lpb:
        nblk = (E + 63) // 64
        blk = blob.reshape(nblk, -1)
        rows = blk[:, -8 * 256:].view(np.uint32).reshape(nblk, 8, 64)   # lo sr sl se sw | hi sr sl se sw
        for k, target in ((0, 0xFFFFFFFF), (1, 0xFFFFFFF0), (2, 0xFFFFFFFF), (3, 0x7FFFFFFB)):
            old = rows[:, k, :].astype(np.int64).reshape(-1)[:E]
            add[:, k] = target - old
            rows[:, k, :] = target
    else:
        tail = 16 if exact is False else 0     # bitboard handles: the two step counters of hexvec_attach_pool come last
        st = blob[len(blob) - tail - E * 32:len(blob) - tail].view(np.int64).reshape(E, 4)
        for k, target in ((0, 2 ** 31 - 1), (1, 2 ** 32 - 16), (2, 2 ** 31 - 1), (3, 2 ** 31 - 5)):
            add[:, k] = target - st[:, k]
            st[:, k] = target
    sd["blob"] = torch.from_numpy(blob).to(sd["blob"].device)
    b.load_state_dict(sd)
    if exact is True:
        for _ in range(T):
            a.step()
            b.step()
    else:
        for _ in range(3):      # carried across launches too
            a.rollout_selfplay(T // 3, out={})
            b.rollout_selfplay(T // 3, out={})
    sa, sb = _np(a.episode_stats()), _np(b.episode_stats())
    assert sa[2] > 5 * E                                        # every env finished episodes after the import
    want = sa + add.sum(0).astype(np.float64)
    assert np.array_equal(sb, want), (sb - want)
    assert sb[1] > E * 2.0 ** 32 and sb[2] > E * 2.0 ** 31       # past what 32-bit sums could hold


def test_completion_ticket_counts_boards_behind_their_outputs(hx):
    """hexvec_set_ticket: every board's wavefront adds 1 to the host-pinned word behind its outputs (the single-board
    mirrors spin on it instead of a stream wait); literal-engine handles only"""
    import ctypes as C
    L = hx._lib.lib()
    n, E = 5, 7
    env = hx.HexVecEnv(n, E, encoding="A", two_sided=True, auto_reset=False, seed=3)
    tk = torch.zeros(1, dtype=torch.int32).pin_memory()
    ret = torch.full((1, E), 0x55, dtype=torch.int8).pin_memory()
    assert L.hexvec_set_ticket(env._h, C.c_void_p(tk.data_ptr())) == 0
    acts = torch.arange(E, dtype=torch.int32, device="cuda").view(1, E).contiguous()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for k in range(1, 4):
        ret.fill_(0x55)
        a = (acts + 5 * (k - 1)).contiguous()
        assert L.hexvec_apply_moves(env._h, C.c_void_p(a.data_ptr()), 1, C.c_void_p(ret.data_ptr()), None, None, None, None, st) == 0
        for _ in range(2000000):
            if int(tk[0]) == k * E:
                break
        assert int(tk[0]) == k * E
        assert bool((ret != 0x55).all())              # the outputs were visible before the last increment
    torch.cuda.synchronize()
    assert L.hexvec_set_ticket(env._h, None) == 0
    fast = hx.HexVecEnv(n, E, encoding="B")
    rc = L.hexvec_set_ticket(fast._h, C.c_void_p(tk.data_ptr()))        # HEXVEC_E_MODE: a throughput handle
    assert rc != 0 and b"literal engine" in L.hexvec_last_error()


def test_rollout_outputs_are_checked_before_the_launch(hx):
    """a rollout writes T rows into the caller's buffers: too short, wrong dtype or a CPU tensor raises in the host
    layer (the library itself sees raw pointers)"""
    env = hx.HexVecEnv(5, 256, encoding="B", seed=1)
    good = env.rollout_selfplay(4)
    assert good["obs"].shape == (4, 256, 25)
    with pytest.raises(ValueError):
        env.rollout_selfplay(8, good)                                    # buffers hold 4 rows
    with pytest.raises(ValueError):
        env.rollout_selfplay(4, dict(good, reward=good["reward"].double()))
    with pytest.raises(ValueError):
        env.rollout_selfplay(4, dict(good, done=good["done"].cpu()))
    with pytest.raises(KeyError):
        env.rollout_selfplay(4, dict(good, winner=good["done"]))
    env.rollout_selfplay(4, {"obs": good["obs"]})                        # a subset is fine
    env.rollout_selfplay(2, good)                                        # so is a larger buffer
