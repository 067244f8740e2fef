"""Pin the CPU oracle (oracle/hex_oracle.c) against fixtures generated from the
live reference (tests/golden/make_fixtures.py).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import hex_oracle as ho

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _unpack(bits, n):
    return np.unpackbits(bits, axis=-1)[..., :n * n]


def _mask(L, g):
    m = np.zeros(g.n * g.n, np.uint8)
    L.hexo_legal_mask(C.byref(g), m.ctypes.data_as(C.POINTER(C.c_uint8)))
    return m


def test_engine_a_matches_reference():
    """HexGame.py:16-142: return value, done, player flip, board, mask, final regions."""
    L, d = ho.lib(), _load("engineA.npz")
    checked = 0
    for n in d["sizes"]:
        for gi in range(int(d[f"N{n}_ngames"])):
            k = f"N{n}_g{gi}_"
            g = ho.Game()
            L.hexo_game_init(C.byref(g), int(n), ho.ENC_A, int(d[k + "first"]), None)
            masks = _unpack(d[k + "mask"], int(n))
            for t, a in enumerate(d[k + "actions"]):
                ret = L.hexo_make_move(C.byref(g), int(a))
                assert ret == d[k + "ret"][t], (k, t)
                assert bool(g.done) == bool(d[k + "done"][t]), (k, t)
                assert g.cur == d[k + "cur_after"][t], (k, t)
                assert np.array_equal(g.board_np().ravel(), d[k + "board"][t]), (k, t)
                assert np.array_equal(_mask(L, g), masks[t]), (k, t)
                checked += 1
            assert np.array_equal(g.regions_np(), d[k + "regions"].astype(np.int32)), k
    assert checked > 1500


def test_env_b_matches_reference():
    """HexSingleGame.py:156-271 incl. illegal plies and plies after done."""
    L, d = ho.lib(), _load("envB.npz")
    checked = 0
    for n in d["sizes"]:
        for gi in range(int(d[f"N{n}_ngames"])):
            k = f"N{n}_g{gi}_"
            e = ho.EnvB()
            e.winner = -1
            L.hexo_envb_reset(C.byref(e), int(n))
            assert np.array_equal(e.sim.board_np().ravel(), d[k + "reset_obs"])
            masks = _unpack(d[k + "mask"], int(n))
            rb = C.c_int()
            for t, a in enumerate(d[k + "actions"]):
                done = L.hexo_envb_step(C.byref(e), int(a), C.byref(rb))
                assert e.winner == d[k + "ret"][t], (k, t)
                assert rb.value == d[k + "r_black"][t], (k, t)
                assert bool(done) == bool(d[k + "done"][t]), (k, t)
                assert e.sim.cur == d[k + "engine_cur_after"][t], (k, t)
                assert np.array_equal(e.sim.board_np().ravel(), d[k + "obs"][t]), (k, t)
                assert np.array_equal(_mask(L, e.sim), masks[t]), (k, t)
                checked += 1
            assert np.array_equal(e.sim.regions_np(), d[k + "regions"].astype(np.int32)), k
    assert checked > 1500


def test_selfplay_matches_reference():
    """SelfplayWrapper.py:69-89, :146-199 with the recorded opponent actions."""
    L, d = ho.lib(), _load("selfplay.npz")
    checked = 0
    for n in d["sizes"]:
        for ei in range(int(d[f"N{n}_neps"])):
            k = f"N{n}_e{ei}_"
            s = ho.SelfPlay()
            s.env.winner = -1
            agent = int(d[k + "agent"])
            L.hexo_selfplay_reset(C.byref(s), int(n), agent, int(d[k + "reset_opp_action"]))
            assert np.array_equal(s.env.sim.board_np().ravel(), d[k + "reset_obs"]), k
            assert np.array_equal(_mask(L, s.env.sim), _unpack(d[k + "reset_mask"], int(n))), k
            masks = _unpack(d[k + "mask"], int(n))
            ar, used = C.c_int(), C.c_int()
            for t, a in enumerate(d[k + "actions"]):
                oa = int(d[k + "opp_actions"][t])
                done = L.hexo_selfplay_step(C.byref(s), int(a), max(oa, 0), C.byref(ar),
                                            C.byref(used))
                assert bool(used.value) == (oa >= 0), (k, t)
                assert ar.value == d[k + "reward"][t], (k, t)
                assert bool(done) == bool(d[k + "done"][t]), (k, t)
                assert np.array_equal(s.env.sim.board_np().ravel(), d[k + "obs"][t]), (k, t)
                assert np.array_equal(_mask(L, s.env.sim), masks[t]), (k, t)
                checked += 1
    assert checked > 500


def test_env_a_matches_reference():
    """HexGame.py:206-295, :332-349 (hex-v0, agent BLACK) with recorded opponent actions."""
    L, d = ho.lib(), _load("envA.npz")
    checked = 0
    for n in d["sizes"]:
        for ei in range(int(d[f"N{n}_neps"])):
            k = f"N{n}_e{ei}_"
            e = ho.EnvA()
            e.winner = -1
            last = C.c_int()
            L.hexo_enva_reset(C.byref(e), int(n), int(d[k + "first"]),
                              max(int(d[k + "reset_opp_action"]), 0), C.byref(last))
            assert last.value == d[k + "reset_last_move_opponent"], k
            assert np.array_equal(e.sim.board_np().ravel(), d[k + "reset_obs"]), k
            assert np.array_equal(_mask(L, e.sim), _unpack(d[k + "reset_mask"], int(n))), k
            masks = _unpack(d[k + "mask"], int(n))
            rew, used = C.c_int(), C.c_int()
            for t, a in enumerate(d[k + "actions"]):
                oa = int(d[k + "opp_actions"][t])
                done = L.hexo_enva_step(C.byref(e), int(a), max(oa, 0), C.byref(rew),
                                        C.byref(last), C.byref(used))
                assert bool(used.value) == (oa >= 0), (k, t)
                assert rew.value == d[k + "reward"][t], (k, t)
                assert bool(done) == bool(d[k + "done"][t]), (k, t)
                assert e.winner == d[k + "winner"][t], (k, t)
                assert last.value == d[k + "last_move_opponent"][t], (k, t)
                assert np.array_equal(e.sim.board_np().ravel(), d[k + "obs"][t]), (k, t)
                assert np.array_equal(_mask(L, e.sim), masks[t]), (k, t)
                checked += 1
    assert checked > 300


def test_preplaced_constructor_replay():
    """HexGame.py:53-61 / HexSingleGame.py:57-65: board -> regions, region_counter."""
    L, d = ho.lib(), _load("preplaced.npz")
    for n in d["sizes"]:
        for gi in range(6):
            k = f"N{n}_g{gi}_"
            for enc, tag in ((ho.ENC_A, "A"), (ho.ENC_B, "B")):
                b = np.ascontiguousarray(d[k + "board" + tag].ravel(), np.int8)
                g = ho.Game()
                L.hexo_game_init(C.byref(g), int(n), enc, 0, b.ctypes.data_as(C.POINTER(C.c_int8)))
                assert np.array_equal(g.regions_np(), d[k + "regions" + tag].astype(np.int32)), (k, tag)
                assert list(g.region_counter) == list(d[k + "counter" + tag]), (k, tag)


def test_winner_is_plain_connectivity():
    """Independent check: the label-merging winner equals BFS connectivity."""
    L = ho.lib()
    rng = np.random.default_rng(5)
    for n in (3, 6, 9, 14):
        for _ in range(20):
            g = ho.Game()
            L.hexo_game_init(C.byref(g), n, ho.ENC_A, 0, None)
            order = rng.permutation(n * n)
            for a in order:
                mover = g.cur
                ret = L.hexo_make_move(C.byref(g), int(a))
                b = g.board_np()
                # BFS for the mover between its two edges
                stones = (b == mover)
                if mover == 1:
                    stones = stones.T
                seen = np.zeros_like(stones)
                stack = [(0, x) for x in range(n) if stones[0, x]]
                for p in stack:
                    seen[p] = True
                win = False
                while stack:
                    y, x = stack.pop()
                    if y == n - 1:
                        win = True
                    for dy, dx in ((-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0)):
                        yy, xx = y + dy, x + dx
                        if 0 <= yy < n and 0 <= xx < n and stones[yy, xx] and not seen[yy, xx]:
                            seen[yy, xx] = True
                            stack.append((yy, xx))
                assert (ret == mover) == win
                if win:
                    break


def test_config1_thousand_games_5x5():
    """BASELINE config 1: 1 000 reference games at 5x5 (python-random policy, seed 0): the oracle
    returns the same make_move code on every ply and ends on the same ply with the same winner"""
    L, d = ho.lib(), _load("config1_5x5.npz")
    n, acts, rets = int(d["n"]), d["actions"], d["ret"]
    wins = [0, 0]
    for g in range(acts.shape[0]):
        sim = ho.Game()
        L.hexo_game_init(C.byref(sim), n, ho.ENC_A, 0, None)
        for t in range(acts.shape[1]):
            if acts[g, t] < 0:
                break
            assert L.hexo_make_move(C.byref(sim), int(acts[g, t])) == rets[g, t], (g, t)
        last = int((acts[g] >= 0).sum()) - 1
        assert sim.done and rets[g, last] in (0, 1)
        wins[int(rets[g, last])] += 1
    assert sum(wins) == 1000 and wins[0] > wins[1]          # the first mover wins more often


@pytest.mark.parametrize("n,enc_flags", [(5, ho.F_ENC_B | ho.F_AGENT_RANDOM), (7, ho.F_ENC_B | ho.F_AGENT_WHITE),
                                         (6, ho.F_FIRST_WHITE), (11, ho.F_ENC_B | ho.F_AGENT_RANDOM)])
def test_vector_w1_rollout_is_the_stepped_contract(n, enc_flags):
    """layer 2 only: hexo_vec_rollout_selfplay (what the GPU's rollout_selfplay is compared with) records exactly
    what hexo_vec_step returns step by step -- same draws, plus the agent's action, which is always a legal cell of
    the position the agent was shown; threads do not change the result"""
    E, T = 96, 3 * n * n
    a, b = ho.OracleVec(n, E, seed=5, flags=enc_flags, env_offset=17), ho.OracleVec(n, E, seed=5, flags=enc_flags, env_offset=17)
    b.set_threads(4)
    prev_legal = a.reset()[1]
    b.reset()
    obs, legal, act, rew, done = b.rollout_selfplay(T)
    for t in range(T):
        so, sl, sr, sd, sw = a.step()
        assert np.array_equal(obs[t], so) and np.array_equal(legal[t], sl), t
        assert np.array_equal(rew[t], sr) and np.array_equal(done[t], sd), t
        assert prev_legal[np.arange(E), act[t]].all(), t
        prev_legal = sl
    assert np.array_equal(a.episode_stats(), b.episode_stats()) and a.episode_stats()[2] > E
    assert np.array_equal(a.get_state()[0], b.get_state()[0])
