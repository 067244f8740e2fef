"""Split step (hexvec_step_agent / hexvec_step_opponent) and the vectorised self-play wrapper."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import hex_oracle as ho  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hx():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    import hex_gym_env_amd
    return hex_gym_env_amd


def _np(t):
    return t.detach().cpu().numpy()


LAYOUTS = pytest.mark.parametrize("lay", [False, True], ids=["bitboard", "lane-per-board"])


@pytest.mark.parametrize("n,enc,agent,first", [(5, "B", None, 0), (11, "B", 1, 0), (7, "B", 0, 0),
                                               (5, "A", None, 1), (9, "A", None, 0), (13, "B", None, 0)])
@LAYOUTS
def test_split_step_equals_fused_step(hx, n, enc, agent, first, lay):
    """step_agent + step_opponent rounds with library-drawn actions == the oracle's whole step
    (on the lane-per-board layout the opponent's view is read out of the agent's image as the negated
    transpose: hexvec_lpb.h emit_gather)"""
    E, K = 192, 5 * n
    g = hx.HexVecEnv(n, E, encoding=enc, agent_player_num=agent, first_player=first, seed=21, rollout_layout=lay)
    f = ho.F_ENC_B | (ho.F_AGENT_RANDOM if agent is None else (ho.F_AGENT_WHITE if agent else 0)) \
        if enc == "B" else (ho.F_FIRST_WHITE if first else 0)
    o = ho.OracleVec(n, E, seed=21, flags=f)
    rng = np.random.default_rng(n)
    for k in range(K):
        acts = None
        if k % 2:   # external agent actions, some of them illegal
            acts = rng.integers(0, n * n, size=E).astype(np.int32)
        g.done.zero_()
        obs, legal, pend = g.step_agent(acts)
        rounds = 0
        while bool(pend.any()):
            obs, rew, done, legal, win, pend = g.step_opponent()
            rounds += 1
        assert rounds <= 2
        oo, ol, orw, od, ow = o.step(acts)
        assert np.array_equal(_np(g.obs).reshape(E, -1), oo), k
        assert np.array_equal(_np(g.legal), ol), k
        assert np.array_equal(_np(g.reward), orw), k
        assert np.array_equal(_np(g.done), od), k
        assert np.array_equal(_np(g.winner), ow), k
    assert np.array_equal(_np(g.episode_stats()), o.episode_stats())
    with pytest.raises(RuntimeError):
        hx.HexVecEnv(n, 8, encoding=enc, two_sided=True).step_agent()


@LAYOUTS
def test_opponent_sees_its_own_view_and_its_reply_lands(hx, lay):
    """a deterministic 'first legal cell' opponent: the reply must be the first empty cell of the
    opponent's view (= -board^T), HexSingleGame.py:265-271 + SelfplayWrapper.py:161"""
    n, E = 6, 256
    g = hx.HexVecEnv(n, E, encoding="B", agent_player_num=0, seed=5, rollout_layout=lay)
    obs, legal = g.reset()
    rng = np.random.default_rng(0)
    for k in range(12):
        before = _np(obs).copy()
        lm = _np(legal)
        acts = np.array([rng.choice(np.flatnonzero(m)) for m in lm], np.int32)
        g.done.zero_()
        o1, l1, pend = g.step_agent(acts)
        o1, l1, pend = _np(o1).copy(), _np(l1).copy(), _np(pend).copy()
        exp_mid = before.reshape(E, -1).copy()
        exp_mid[np.arange(E), acts] = -1
        exp_mid = -np.transpose(exp_mid.reshape(E, n, n), (0, 2, 1))      # the opponent's view
        live = pend == 1
        assert np.array_equal(o1[live], exp_mid[live])
        assert np.array_equal(l1[live], (exp_mid[live].reshape(-1, n * n) == 0))
        opp = np.array([np.flatnonzero(m)[0] if m.any() else 0 for m in l1], np.int32)
        obs, rew, done, legal, win, pend2 = g.step_opponent(opp)
        assert not bool(pend2.any())                   # BLACK agents: nobody opens
        after = _np(obs)
        exp = exp_mid.reshape(E, -1).copy()
        exp[np.arange(E), opp] = -1
        exp = -np.transpose(exp.reshape(E, n, n), (0, 2, 1))
        cont = live & (_np(done) == 0)
        assert np.array_equal(after[cont], exp[cont])


def test_vec_selfplay_env_with_policy_buffer(hx):
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, RandomLegalPolicy

    class FirstLegal:
        calls = 0

        def choose_actions(self, obs, legal):
            FirstLegal.calls += 1
            return legal.float().argmax(1)

    n, E = 5, 512
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=4, seed=3)
    assert len(env.get_opponent_models()) == 4 and isinstance(env.best_model, RandomLegalPolicy)
    env.set_opponent_model(2, FirstLegal(), 0.7)
    assert env.best_score == 0.7 and isinstance(env.best_model, FirstLegal)
    obs, legal = env.reset()
    col = _np(env.vec.agent_colours())
    stones = (_np(obs) != 0).reshape(E, -1).sum(1)
    assert np.array_equal(stones, col)                 # WHITE agents face one opening stone
    # boards whose opponent is FirstLegal opened at the first cell of ITS view = cell 0 -> agent sees it at 0
    is_fl = (_np(env.slot) == -1) | (_np(env.slot) == 2)
    w = (col == 1) & is_fl
    assert w.any() and (_np(obs).reshape(E, -1)[w, 0] == 1).all()
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = 0
    for k in range(40):
        a = torch.multinomial(env.legal_actions().float(), 1, generator=g).squeeze(1)
        obs, rew, done, legal, win = env.step(a)
        assert bool(((obs.reshape(E, -1) == 0) == (legal != 0)).all())
        tot += int(done.sum())
    st = _np(env.episode_stats())
    assert st[2] == tot and tot > E
    assert FirstLegal.calls > 40
    # ~80 % of the boards use the best model, the rest the buffer
    frac_best = float((env.slot == -1).float().mean())
    assert 0.7 < frac_best < 0.9
    env.set_eval(True)
    env.reset()
    assert sorted(set(_np(env.slot).tolist())) == [0, 1, 2, 3]


def test_reference_trained_actor_beats_random_play(hx):
    """End-to-end check of the observation conventions: the actor of the reference's shipped 5x5
    checkpoint (weights exported as data by tests/golden/make_fixtures.py), used as the OPPONENT
    through the split step, must beat a uniform-random agent most of the time; as the AGENT against
    random opponents it must win most games.  A wrong perspective / transpose makes it play badly."""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
    n, E = 5, 4096
    actor = MlpActorPolicy.from_npz(os.path.join(GOLDEN, "sb3_5x5_actor.npz")).to("cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=2, base_model=actor, seed=8)
    obs, legal = env.reset()
    for _ in range(60):
        a = torch.multinomial(legal.float(), 1, generator=g).squeeze(1)
        obs, rew, done, legal, win = env.step(a)
    st = _np(env.episode_stats())
    random_agent_winrate = st[3] / st[2]
    env2 = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=2, seed=9)
    obs, legal = env2.reset()
    for _ in range(60):
        obs, rew, done, legal, win = env2.step(actor.choose_actions(obs, legal))
    st2 = _np(env2.episode_stats())
    trained_agent_winrate = st2[3] / st2[2]
    assert st[2] > 5 * E and st2[2] > 5 * E
    assert random_agent_winrate < 0.25, random_agent_winrate
    assert trained_agent_winrate > 0.75, trained_agent_winrate


@pytest.mark.parametrize("exact", [False, True, "lpb"])
def test_state_dict_roundtrip_and_render(hx, exact):
    n, E = 7, 200
    kw = dict(rollout_layout=True) if exact == "lpb" else dict(exact_labels=exact)
    a = hx.HexVecEnv(n, E, encoding="B", seed=2, **kw)
    for _ in range(9):
        a.step()
    sd = a.state_dict()
    ref = [tuple(_np(x).copy() for x in a.step()) for _ in range(15)]
    b = hx.HexVecEnv(n, E, encoding="B", seed=2, **kw)
    b.load_state_dict(sd)
    for r in ref:
        out = b.step()
        for x, y in zip(out, r):
            assert np.array_equal(_np(x), y)
    assert np.array_equal(_np(a.episode_stats()), _np(b.episode_stats()))
    with pytest.raises(ValueError):
        hx.HexVecEnv(n, E + 1, encoding="B").load_state_dict(sd)
    txt = b.render([0, 3])
    assert txt.count("\n") == 2 * (2 + 2 * n) + 1 and "O" in txt


def test_device_board_sampler_matches_reference_distribution():
    """sample_boards vs 4000 draws of the reference's random_board (fixture): same support, close
    histograms of stone count and occupied rows / columns"""
    from hex_gym_env_amd.vec_selfplay import sample_boards
    d = np.load(os.path.join(GOLDEN, "sample_board.npz"))
    g = torch.Generator(device="cuda").manual_seed(4)
    for n in d["sizes"]:
        n = int(n)
        M = 40000
        b = _np(sample_boards(n, M, g, "cuda"))
        assert ((b == -1).reshape(M, -1).sum(1) == (b == 1).reshape(M, -1).sum(1)).all()
        cnt = (b != 0).reshape(M, -1).sum(1)
        rows, cols = (b != 0).any(2).sum(1), (b != 0).any(1).sum(1)
        assert (cnt % 2 == 0).all() and cnt.max() <= (n - 2) ** 2 and max(rows.max(), cols.max()) <= n - 2
        for mine, key in ((cnt, "stones"), (rows, "rows"), (cols, "cols")):
            ref = d[f"N{n}_stats_{key}_hist"].astype(float)
            h = np.bincount(mine, minlength=len(ref)).astype(float)
            assert len(h) == len(ref)
            assert np.abs(h / M - ref / ref.sum()).max() < 0.03, (n, key)


@pytest.mark.parametrize("n,enc", [(5, "B"), (11, "B"), (7, "A"), (14, "B")])
@pytest.mark.parametrize("lay", [False, True], ids=["bitboard", "lane-per-board"])
def test_set_boards_matches_exact_engine(hx, n, enc, lay):
    """start positions on the throughput layouts == constructor replay on the exact engine
    (which tests/test_gpu_mirror.py pins to the reference's regions)"""
    from hex_gym_env_amd.vec_selfplay import sample_boards
    E = 192
    g = torch.Generator(device="cuda").manual_seed(n)
    boards = sample_boards(n, E, g, "cuda")
    if enc == "A":
        boards = torch.where(boards == -1, 0, torch.where(boards == 1, 1, 2)).to(torch.int8)
    fast = hx.HexVecEnv(n, E, encoding=enc, two_sided=True, seed=1, rollout_layout=lay)
    exact = hx.HexVecEnv(n, E, encoding=enc, two_sided=True, seed=1, exact_labels=True)
    obs, legal, pend = fast.set_boards(boards)
    exact.set_state(boards.reshape(E, -1))
    assert torch.equal(obs, boards) and not bool(pend.any())
    assert torch.equal(fast.get_state()[0], exact.get_state()[0])
    empty = 2 if enc == "A" else 0
    for k in range(2 * n):
        a = torch.multinomial(fast.legal.float(), 1, generator=g).squeeze(1)
        fo = [x.clone() for x in fast.step(a)]
        eo = exact.step(a)
        for x, y in zip(fo, eo):
            assert torch.equal(x, y), k
        assert torch.equal(fast.legal, (fast.obs.reshape(E, -1) == empty).to(torch.uint8))
    # a masked call touches only the masked boards
    m = torch.zeros(E, dtype=torch.uint8, device="cuda")
    m[::3] = 1
    before = fast.get_state()[0].clone()
    fast.set_boards(boards, m)
    after = fast.get_state()[0]
    assert torch.equal(after[m == 0], before[m == 0])
    assert torch.equal(after[m != 0], boards.reshape(E, -1)[m != 0])


def test_vec_selfplay_with_sampled_start_positions(hx):
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv
    n, E = 7, 1024
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=3, seed=6, sample_board=True)
    obs, legal = env.reset()
    o = _np(obs).reshape(E, -1)
    col = _np(env.vec.agent_colours())
    own, opp = (o == -1).sum(1), (o == 1).sum(1)
    assert (own[col == 0] == opp[col == 0]).all()          # BLACK agents: balanced start position
    assert (opp[col == 1] == own[col == 1] + 1).all()      # WHITE agents: the opponent has opened
    assert (own + opp).mean() > 2                          # positions are not empty
    assert np.array_equal(_np(legal), (o == 0).astype(np.uint8))
    g = torch.Generator(device="cuda").manual_seed(2)
    tot = 0
    for k in range(30):
        a = torch.multinomial(env.legal_actions().float(), 1, generator=g).squeeze(1)
        obs, rew, done, legal, win = env.step(a)
        o = _np(obs).reshape(E, -1)
        d = _np(done).astype(bool)
        tot += int(d.sum())
        assert np.array_equal(_np(legal), (o == 0).astype(np.uint8))
        own, opp = (o == -1).sum(1), (o == 1).sum(1)
        assert (own[col == 0] == opp[col == 0]).all() and (opp[col == 1] == own[col == 1] + 1).all()
        assert not bool(env.vec.pending.any())
    assert tot > E // 2 and _np(env.episode_stats())[2] == tot


@pytest.mark.parametrize("n,E", [(7, 32768), (13, 16384)])
def test_baseline_config_sizes_selfplay_with_mlp_policy(hx, n, E):
    """BASELINE.json configs[2] (7x7, 32 768 envs) and configs[4] (13x13, 131 072 / 8 GPUs):
    SelfplayWrapper semantics with an MLP policy as the agent (SB3 default 64-64 tanh net, random
    init) and a buffer of MLP opponents; size-independent properties + the stats checksum"""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
    torch.manual_seed(0)
    agent = MlpActorPolicy(n * n).to("cuda")
    opp = MlpActorPolicy(n * n).to("cuda")
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=4, base_model=opp, seed=12)
    obs, legal = env.reset()
    col = env.vec.agent_colours()
    tot = 0
    for k in range(25):
        a = agent.choose_actions(obs, legal)
        assert bool(torch.gather(legal, 1, a.view(-1, 1)).all())          # the mask keeps it legal
        obs, rew, done, legal, win = env.step(a)
        o = obs.reshape(E, -1)
        assert torch.equal(legal, (o == 0).to(torch.uint8))
        own, other = (o == -1).sum(1), (o == 1).sum(1)
        assert torch.equal(own, torch.where(col == 0, other, other - 1))  # it is always the agent's turn
        assert bool(((rew != 0) <= done.bool()).all())
        assert bool(((win == col.to(torch.int8)) == (rew > 0))[done.bool()].all())
        tot += int(done.sum())
    st = _np(env.episode_stats())
    assert st[2] == tot and st[3] == (st[0] + st[2]) / 2                  # wins - losses == sum of returns


def test_sb3_adapter_with_opponent_buffer_and_callback_surface(hx):
    """the calls of minihex/EvaluationCallback.py:21-56 against the vectorised env, with SB3-style
    models (predict(obs, deterministic, action_masks)) as buffer entries"""
    from hex_gym_env_amd.sb3_adapter import HexSB3VecEnv

    class FakeSB3Model:
        n_calls = 0

        def predict(self, obs, deterministic=False, action_masks=None):
            FakeSB3Model.n_calls += 1
            assert obs.ndim == 3 and action_masks.shape == (obs.shape[0], obs.shape[1] * obs.shape[2])
            return np.array([np.flatnonzero(m)[-1] for m in action_masks]), None     # last legal cell

        def save(self, path):
            self.saved = path

    n, E, K = 5, 128, 4
    venv = HexSB3VecEnv(board_size=n, num_envs=E, buffer_size=K, scores=np.zeros(K), seed=1)
    obs = venv.reset()
    model = FakeSB3Model()
    # --- the callback's sequence
    venv.set_eval(True)
    venv.reset()
    venv.set_eval(False)
    scores = venv.get_scores()
    assert len(venv.get_opponent_models()) == K and scores.shape == (K,)
    worst = np.arange(K)[scores == np.min(scores)]
    venv.set_opponent_model(int(worst[0]), model, 0.9)
    assert venv.best_score == 0.9 and venv.get_scores()[worst[0]] == 0.9
    venv.save_best_model()
    assert model.saved.startswith("models/best_model_")
    # --- the new best model now answers ~80 % of the boards: its stones sit at the LAST cell of
    #     the opponent's view, i.e. at cell (N-1,N-1) transposed = the agent's last cell
    obs = venv.reset()
    rng = np.random.default_rng(0)
    ends = 0
    for k in range(30):
        masks = np.stack(venv.env_method("action_masks"))
        acts = np.array([rng.choice(np.flatnonzero(m)) for m in masks])
        obs, rew, done, infos = venv.step(acts)
        ends += int(done.sum())
        for i in np.flatnonzero(done):
            assert "terminal_observation" in infos[i] and infos[i]["episode"]["l"] >= 1
    assert FakeSB3Model.n_calls >= 30 and ends > 0
    with pytest.raises(AttributeError):
        HexSB3VecEnv(board_size=n, num_envs=8).set_eval(True)
    venv.close()


@LAYOUTS
def test_prewon_start_position_is_noticed_on_the_next_ply(hx, lay):
    """quirk q9: a start position in which a colour is already connected ends the game on that
    colour's next ply wherever it is played (the reference tests regions[cur][-1,-1] after every move
    of the mover, HexGame.py:101-105) -- fast path == exact engine"""
    n, E = 5, 64
    b = torch.zeros((E, n, n), dtype=torch.int8, device="cuda")
    b[:, :, 2] = -1                      # BLACK column: top-bottom connected
    b[:, 0, 0] = 1
    b[:, 4, 4] = 1                       # some WHITE stones, not connected
    fast = hx.HexVecEnv(n, E, encoding="B", two_sided=True, seed=1, auto_reset=True, rollout_layout=lay)
    exact = hx.HexVecEnv(n, E, encoding="B", two_sided=True, seed=1, exact_labels=True)
    fast.set_boards(b)
    exact.set_state(b.reshape(E, -1))
    a = torch.full((E,), 1, dtype=torch.int32, device="cuda")       # BLACK plays a cell away from the column
    fo = [x.clone() for x in fast.step(a)]
    eo = exact.step(a)
    for x, y in zip(fo, eo):
        assert torch.equal(x, y)
    assert bool(fo[2].all()) and bool((fo[1] == 1).all())          # done, reward +1 for BLACK


def test_entry_points_reject_wrong_handle_modes(hx):
    fast2 = hx.HexVecEnv(5, 8, encoding="B", two_sided=True)
    fastsp = hx.HexVecEnv(5, 8, encoding="B")
    exact = hx.HexVecEnv(5, 8, encoding="B", two_sided=True, auto_reset=False)
    z = np.zeros((3, 8), np.int32)
    for fn in (lambda: fast2.apply_moves(z),                       # literal plies need an exact handle
               lambda: fastsp.rollout_random(4),                   # two-sided only
               lambda: fast2.rollout_selfplay(4),                  # self-play only
               lambda: exact.rollout_random(4),                    # needs auto reset
               lambda: exact.step_agent(),                         # split step: fast self-play handles
               lambda: fast2.step_agent(),
               lambda: exact.set_boards(np.zeros((8, 5, 5), np.int8)),
               lambda: fastsp.set_state(np.zeros((8, 25), np.int8))):
        with pytest.raises(RuntimeError):
            fn()
    assert fastsp.get_state()[1] is None and exact.get_state()[1] is not None
    assert fastsp.rollout_selfplay(0, out={}) == {}                # T = 0 is a no-op


def test_steps_can_be_captured_in_a_hip_graph(hx):
    """the ABI enqueues on the caller's stream and never synchronises, so a launch-bound loop of
    steps can be captured once (torch.cuda.CUDAGraph = hipGraph) and replayed; results == oracle"""
    n, E, K = 7, 128, 8
    env = hx.HexVecEnv(n, E, encoding="B", seed=3)
    ref = ho.OracleVec(n, E, seed=3, flags=ho.F_ENC_B | ho.F_AGENT_RANDOM)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        env.step()                                   # warm-up outside the capture
        ref.step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(K):
                env.step()                           # K device-drawn self-play steps per replay
    for rep in range(5):
        graph.replay()
        torch.cuda.synchronize()
        for _ in range(K):
            oo, ol, orw, od, ow = ref.step()
        assert np.array_equal(_np(env.obs).reshape(E, -1), oo), rep
        assert np.array_equal(_np(env.reward), orw) and np.array_equal(_np(env.done), od), rep
    assert np.array_equal(_np(env.episode_stats()), ref.episode_stats())


def test_bench_json_contract():
    """bench.py prints ONE JSON line with the contract's keys, a roofline and a cpu_baseline object"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "10",
                          "--envs", "8192", "--chunk", "20", "--cpu-seconds", "1.5", "--layout", "rollout"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 10 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "u8"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["timed_reps"] >= 20 and d["warmup_internal_ms"] >= 100.0
    assert d["reps_per_event"] >= 1 and d["timed_reps"] % d["reps_per_event"] == 0   # one HIP event per group of repetitions
    assert abs(d["wall_ms_per_step"] - d["ms_per_step"]) / d["ms_per_step"] < 0.5   # launch-bound at this size
    assert abs(r["obs_only_frac"] - r["frac"] * 121 / 251) < 1e-9 and r["kernel"].startswith("k_lpb") and d["host_stepped"]["kernel"].startswith("k_qpb")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    cs = d["collect_step"]                  # BASELINE config 3 with the policy in the loop (secondary entry)
    assert "error" not in cs and cs["us_per_step"] > 0 and cs["degenerate_rows"] == 0
    assert abs(cs["agent_steps_per_s"] - 32768 / (cs["us_per_step"] * 1e-6)) / cs["agent_steps_per_s"] < 1e-9
    # the timed kernel at the timed geometry against the oracle, after the timed region (exit code 1 on a mismatch)
    v = d["verified"]
    assert v["equal"] is True and v["all_ranks_equal"] is True and v["envs"] == 2048 and v["steps"] == 20
    assert v["kernel"] == r["kernel"] and set(v["rows"]) == {"obs", "legal", "actions", "reward", "done"}
    # every secondary leg: own warm-up, >= 20 repetitions, median and min / max
    cr = d["collect_rollout"]               # the same collection, 8 steps per graph, rows straight into the rollout buffer
    assert cr["steps_per_graph"] == 8 and 0 < cr["us_per_step"] <= cs["us_per_step"] * 1.05 and cr["degenerate_rows"] == 0
    for leg, key in (("host_stepped", "us_min_max"), ("rollout_w2", "launch_us_min_max"), ("collect_step", "us_min_max"),
                     ("collect_rollout", "us_min_max"), ("ppo_update", "us_min_max")):
        assert d[leg]["reps"] >= 20 and d[leg][key][0] <= d[leg][key][1], leg
    assert d["ppo_update"]["us_per_minibatch"] > 0 and "error" not in d["ppo_update"]


def _bench_under_launcher(nranks):
    """bench.py as the driver starts it for N > 1: one process per GPU under torch.distributed.run,
    RCCL ("nccl") process group; the launcher is a child process, so it starts before any GPU call
    of its own"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + 17 * nranks),
           os.path.join(root, "bench.py"), "--gpus", str(nranks), "--steps", "20", "--warmup", "5",
           "--envs", "8192", "--no-cpu", "--no-host-stepped", "--no-w2", "--no-collect"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == nranks and d["config"]["global_envs"] == 8192 * nranks
    assert d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - nranks * 8192 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    return d


def test_bench_under_torch_distributed_run_one_rank():
    """the N = 1 case of the multi-GPU launch path: RCCL initialises, the stats all_gather and the
    max-over-ranks all_reduce run on the device"""
    d = _bench_under_launcher(1)
    assert d["episodes"]["n"] > 0


def test_bench_starts_its_own_rank_on_the_gpu_box():
    """`bench.py --gpus N` without a launcher starts the ranks itself (torch.distributed.run as a child process, decided
    before anything touches the GPU) and relays rank 0's line; with one GPU: `--gpus 1 --force-launcher`"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-launcher", "--steps", "20",
                          "--warmup", "5", "--envs", "8192", "--layout", "rollout", "--no-cpu", "--no-host-stepped",
                          "--no-w2", "--no-collect"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["global_envs"] == 8192 and d["verified"]["equal"] is True
    if torch.cuda.device_count() < 2:       # and a job this box cannot hold is refused, not reported under another name
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20"],
                             capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert out.returncode != 0 and "n_gpus" not in out.stdout


def test_bench_two_ranks_over_rccl():
    """two ranks, two GPUs (skipped on a one-GPU box; torch.cuda.device_count() does not touch the GPU)"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    d = _bench_under_launcher(2)
    assert d["episodes"]["n"] > 0


def test_collect_step_as_one_hip_graph(hx):
    """policy forward + env step + opponent networks captured once and replayed; invariants hold.
    Two DISTINCT opponent networks are in play (pool vs best), so the slot-dependent member masks
    are part of the replayed graph; no sampling row is ever degenerate and no step ends illegally."""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
    n, E = 7, 2048
    torch.manual_seed(0)
    agent, opp, opp2 = (MlpActorPolicy(n * n).to("cuda") for _ in range(3))
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=5, base_model=opp, seed=4)
    env.set_opponent_model(2, opp2, 0.5)            # pool slot 2 AND best model: a second policy group
    assert env.best_model is opp2 and len(env._groups()) == 2
    env.reset()
    col = env.vec.agent_colours()
    run = env.capture_collect_step(agent)
    before = _np(env.episode_stats())[2]
    tot = 0
    for k in range(40):
        obs, rew, done, legal, win, act = run()
        o = obs.reshape(E, -1)
        assert torch.equal(legal, (o == 0).to(torch.uint8))
        own, other = (o == -1).sum(1), (o == 1).sum(1)
        assert torch.equal(own, torch.where(col == 0, other, other - 1))
        assert int((win == 3).sum()) == 0, k            # nobody ever played an occupied cell
        tot += int(done.sum())
    assert tot > E and _np(env.episode_stats())[2] - before == tot
    assert env.degenerate_rows(agent) == 0 and run.captures == 1
    # the packed weights and the cached opening logits are static buffers of the graph; run() refreshes them when
    # the live networks change in place (training between collection phases): same graph, new behaviour
    with torch.no_grad():
        agent.action_net.bias[5] += 1000.0              # the agent now plays cell 5 whenever it is empty
        for net in (opp, opp2):
            net.action_net.bias[n * n - 1] += 1000.0    # ... and every opponent opens in the last cell
    seen_open = 0
    for k in range(12):
        prev_legal = env.vec.legal.clone()
        obs, rew, done, legal, win, act = run()
        can = prev_legal[:, 5] != 0
        assert bool((act[can] == 5).all()), k
        o = obs.reshape(E, -1)
        opened = (col != 0) & ((o != 0).sum(1) == 1)    # a fresh board on which only the opponent has played
        assert bool((o[opened][:, n * n - 1] == 1).all())
        seen_open += int(opened.sum())
    assert run.captures == 1 and seen_open > 0


def test_captured_collect_step_follows_the_opponent_pool(hx):
    """set_opponent_model / set_eval after the capture (the SelfPlayCallback cycle,
    SelfplayWrapper.py:113-136): the next run() captures again and plays the NEW opponents"""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, RandomLegalPolicy

    class Corner:                                   # always answers with the first / last empty cell
        def __init__(self, last):
            self.last, self.calls = last, 0

        def choose_actions(self, obs, legal):
            self.calls += 1
            idx = torch.arange(legal.shape[1], device=legal.device).expand_as(legal)
            key = torch.where(legal != 0, idx, torch.full_like(idx, -1 if self.last else 10 ** 6))
            return key.argmax(1) if self.last else key.argmin(1)

    n, E = 5, 256
    first, last = Corner(False), Corner(True)
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=3, base_model=first, agent_player_num=0, seed=9)
    env.reset()
    agent = RandomLegalPolicy(env._gen)
    run = env.capture_collect_step(agent)
    def opp_positions(obs):         # cells of the opponent's stones (+1 for a BLACK agent) in ITS row-major order
        o = obs.reshape(E, n, n).transpose(1, 2).reshape(E, -1) == 1
        idx = torch.arange(n * n, device=o.device).expand_as(o)
        return (torch.where(o, idx, torch.full_like(idx, n * n)).amin(1),
                torch.where(o, idx, torch.full_like(idx, -1)).amax(1))

    lo, hi = opp_positions(run()[0])            # 3 warm-up steps + 1 replay: <= 8 stones on a board
    assert bool((hi >= 0).all()) and bool((hi < 8).all())       # "first empty cell" opponents everywhere
    for k in range(3):
        env.set_opponent_model(k, last, 1.0 + k)    # replaces the pool and the best model
    calls = last.calls
    env.reset()
    lo, hi = opp_positions(run()[0])
    assert run.captures == 2 and last.calls > calls
    assert bool((lo >= n * n - 8).all()) and bool((lo < n * n).all())   # "last empty cell" opponents now
    env.set_eval(True)
    run()
    assert run.captures == 3
    env.set_eval(False)
    run()
    assert run.captures == 4 and env.degenerate_rows(agent) == 0


def test_default_random_pool_is_one_policy_group(hx):
    """base_model=None (the reference's default, SelfplayWrapper.py:58-60): buffer_size + 1 random
    policies, identical and stateless -> ONE batched call per opponent round"""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv
    env = VecSelfPlayEnv(board_size=5, num_envs=64, buffer_size=20, seed=1)
    assert len(env._groups()) == 1
    env.reset()
    for _ in range(30):
        a = env.best_model.choose_actions(env.vec.obs, env.vec.legal)
        env.step(a)
    assert env.degenerate_rows() == 0
    # the counter does count: a row without any legal cell, and a NaN logit on a legal cell
    from hex_gym_env_amd.vec_selfplay import MlpActorPolicy
    legal = torch.ones((8, 25), dtype=torch.uint8, device="cuda")
    legal[3] = 0
    env.best_model.choose_actions(env.vec.obs[:8], legal)
    assert env.degenerate_rows() == 1
    mlp = MlpActorPolicy(25).to("cuda")
    with torch.no_grad():
        mlp.action_net.bias[7] = float("nan")
    mlp.choose_actions(env.vec.obs[:8], torch.ones_like(legal))
    assert env.degenerate_rows(mlp) == 1 + 8


def test_fused_masked_sampler_distribution_and_counter(hx):
    """hexvec_sample_masked (one launch): never an illegal cell, frequencies == softmax over the legal cells
    (and == the torch spelling's), uniform with logits None, fresh draws on every call and every graph
    replay, degenerate rows counted"""
    from hex_gym_env_amd.vec_selfplay import sample_masked, sample_masked_torch
    torch.manual_seed(0)
    R, Cc = 200000, 49
    row = torch.randn(Cc, device="cuda") * 1.5
    mask = (torch.rand(Cc, device="cuda") < 0.6).to(torch.uint8)
    mask[3] = 1
    logits, legal = row.expand(R, Cc).contiguous(), mask.expand(R, Cc).contiguous()
    gen = torch.Generator(device="cuda").manual_seed(7)
    bad = torch.zeros((), dtype=torch.long, device="cuda")
    a = sample_masked(logits, legal, gen, bad)
    assert a.dtype == torch.int32 and a.shape == (R,) and int(bad) == 0
    assert bool(mask[a.long()].all())                              # only legal cells
    p = torch.softmax(row.masked_fill(mask == 0, float("-inf")), 0)
    f = torch.bincount(a.long(), minlength=Cc).double() / R
    assert float((f - p.double()).abs().max()) < 4e-3, float((f - p.double()).abs().max())
    ft = torch.bincount(sample_masked_torch(logits, legal), minlength=Cc).double() / R
    assert float((f - ft).abs().max()) < 6e-3
    b = sample_masked(logits, legal, gen, bad)                      # the counter moved: new draws
    assert float((a != b).double().mean()) > 0.5
    u = sample_masked(None, legal, gen, bad)                        # uniform over the legal cells
    fu = torch.bincount(u.long(), minlength=Cc).double() / R
    assert float((fu - mask.double() / mask.sum()).abs().max()) < 4e-3
    # under graph replay the draws keep changing (the counter lives on the device)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        sample_masked(logits, legal, gen, bad)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = sample_masked(logits, legal, gen, bad)
    g.replay()
    torch.cuda.synchronize()
    first = out.clone()
    g.replay()
    torch.cuda.synchronize()
    assert float((first != out).double().mean()) > 0.5 and bool(mask[out.long()].all())
    # rows that cannot be sampled: counted, action 0
    legal2 = legal[:64].clone()
    legal2[5] = 0
    lg2 = logits[:64].clone()
    lg2[9, 3] = float("nan")
    c = sample_masked(lg2, legal2, gen, bad)
    assert int(bad) == 2 and int(c[5]) == 0 and int(c[9]) == 0
    # boards to skip, a logits row shared by every board, actions written in place
    act = torch.zeros(R, dtype=torch.uint8, device="cuda")
    act[::2] = 1
    dst = torch.full((R,), -7, dtype=torch.int32, device="cuda")
    r = sample_masked(row.view(1, Cc), legal, gen, bad, row_active=act, out=dst)
    assert r.data_ptr() == dst.data_ptr() and bool((dst[1::2] == 0).all()) and bool(mask[dst[::2].long()].all())
    fs = torch.bincount(dst[::2].long(), minlength=Cc).double() / (R // 2)
    assert float((fs - p.double()).abs().max()) < 6e-3 and int(bad) == 2      # nothing new was counted
    # odd shapes: one row, 169 columns
    one = sample_masked(torch.zeros((1, 169), device="cuda"), torch.ones((1, 169), dtype=torch.uint8, device="cuda"), gen, bad)
    assert 0 <= int(one[0]) < 169


def test_pick_opponents_kernel_selects_the_right_boards(hx):
    """hexvec_pick_opponents (setup_opponents, SelfplayWrapper.py:97-104, fused with the selection mask): which
    boards get a new slot at each point of the vectorised step, the 80/20 split, untouched boards keep theirs"""
    import ctypes as C
    L = hx._lib.lib()
    E, nbuf = 100000, 7
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    done = (torch.rand(E, generator=g, device=dev) < 0.5).to(torch.uint8)
    pending = torch.randint(0, 3, (E,), generator=g, device=dev).to(torch.uint8)
    white = (torch.rand(E, generator=g, device=dev) < 0.5).to(torch.uint8)
    ended = torch.full((E,), 9, dtype=torch.uint8, device=dev)
    state = torch.zeros(2, dtype=torch.int64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())

    def pick(when):
        slot = torch.full((E,), -5, dtype=torch.int64, device=dev)
        hx._lib.check(L.hexvec_pick_opponents(p(slot), when, p(done), p(pending), p(ended), p(white), E, C.c_float(0.8),
                                              nbuf, C.c_uint64(11), 0, 1, p(state), None))
        torch.cuda.synchronize()
        return slot

    s0 = pick(0)
    assert torch.equal(ended, done)                                            # recorded for `when` = 2
    sel0 = (done != 0) & (pending == 0)
    assert bool((s0[~sel0] == -5).all()) and bool(((s0[sel0] >= -1) & (s0[sel0] < nbuf)).all())
    assert 0.78 < float((s0[sel0] == -1).double().mean()) < 0.82               # 80 % the best model
    members = s0[sel0][s0[sel0] >= 0]
    f = torch.bincount(members, minlength=nbuf).double() / members.numel()
    assert float((f - 1.0 / nbuf).abs().max()) < 0.02                          # the rest uniform over the pool
    s1 = pick(1)
    sel1 = pending == 2
    assert bool((s1[~sel1] == -5).all()) and bool((s1[sel1] >= -1).all())
    done2 = done | (torch.rand(E, generator=g, device=dev) < 0.3).to(torch.uint8)    # more boards ended since
    done_saved, done = done, done2
    s2 = pick(2)
    sel2 = (done2 != 0) & (ended == 0) & (white == 0)
    assert bool(sel2.any()) and bool((s2[~sel2] == -5).all()) and bool((s2[sel2] >= -1).all())
    assert int(state[0]) == 3                                                  # one draw counter step per call
    assert L.hexvec_pick_opponents(p(s2), 5, p(done), p(pending), p(ended), p(white), E, C.c_float(0.8), nbuf,
                                   C.c_uint64(11), 0, 1, p(state), None) == -1       # bad `when`
    # when = 3 / 4: two points of the step in one launch == the two launches under salts s and s + 1
    def pick_salted(when, salt, into):
        hx._lib.check(L.hexvec_pick_opponents(p(into), when, p(done), p(pending), p(ended), p(white), E, C.c_float(0.8),
                                              nbuf, C.c_uint64(11), salt, 0, p(state), None))
    pending[(white == 0) & (pending == 2)] = 0          # as in a real step: only a WHITE agent's board awaits an opening
    for fused, first, second in ((3, 0, 1), (4, 1, 2)):
        done = done_saved if fused == 3 else done2
        if fused == 4:                                  # `ended` as the agent's call of this step left it
            ended.copy_(done_saved)
        two = torch.full((E,), -5, dtype=torch.int64, device=dev)
        pick_salted(first, 20, two)
        pick_salted(second, 21, two)
        if fused == 4:
            ended.copy_(done_saved)
        one = torch.full((E,), -5, dtype=torch.int64, device=dev)
        pick_salted(fused, 20, one)
        torch.cuda.synchronize()
        assert torch.equal(one, two) and bool((one != -5).any()), fused
    done = done2
    # several calls may share a counter under different salts and advance it once
    before = int(state[0])
    slot_a = torch.full((E,), -5, dtype=torch.int64, device=dev)
    slot_b = torch.full((E,), -5, dtype=torch.int64, device=dev)
    hx._lib.check(L.hexvec_pick_opponents(p(slot_a), 1, p(done), p(pending), p(ended), p(white), E, C.c_float(0.5), nbuf,
                                          C.c_uint64(11), 1, 0, p(state), None))
    hx._lib.check(L.hexvec_pick_opponents(p(slot_b), 1, p(done), p(pending), p(ended), p(white), E, C.c_float(0.5), nbuf,
                                          C.c_uint64(11), 2, 1, p(state), None))
    torch.cuda.synchronize()
    assert int(state[0]) == before + 1 and float((slot_a[sel1] != slot_b[sel1]).double().mean()) > 0.3


def _random_position(rows, cells, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    obs = (torch.randint(0, 3, (rows, cells), generator=g, device="cuda") - 1).to(torch.int8)
    legal = (obs == 0).to(torch.uint8)
    return obs, legal


@pytest.mark.parametrize("n,hidden,rows", [(5, (64, 64), 1000), (7, (64, 64), 4096), (11, (64, 64), 777),
                                           (14, (64, 64), 500), (19, (64, 64), 200), (4, (64, 64), 33),      # matrix cores
                                           (2, (64, 64), 70), (3, (64, 64), 2100), (9, (64, 64), 40000),
                                           (5, (64, 64), 60000), (11, (64, 64), 66000), (19, (64, 64), 52000),  # one wavefront per tile (19x19: 92 KB of LDS)
                                           (13, (32, 48), 300), (19, (128, 64), 130), (4, (16, 16), 65),     # general widths
                                           (7, (64, 32), 640)])
def test_fused_actor_matches_fp32_torch_forward(hx, n, hidden, rows):
    """hexvec_mlp_actions (forward + mask + sample in one launch) against the plain torch fp32 forward of the
    same network: logits within 2e-5 absolute (fp32 FMA order + the kernel's exp-based tanh, |err| < 2e-7 per
    activation), the sampled actions exactly those hexvec_sample_masked draws from the kernel's own logits,
    arg-max actions exactly torch's on those logits; skipped and degenerate boards as in the sampler"""
    from hex_gym_env_amd.vec_selfplay import MlpActorPolicy, mlp_actions, sample_masked
    torch.manual_seed(n)
    C_ = n * n
    pol = MlpActorPolicy(C_, hidden).cuda()
    with torch.no_grad():                               # livelier than the default init: tanh well off its linear range
        for p_ in pol.parameters():
            p_.mul_(3.0)
    obs, legal = _random_position(rows, C_, 5)
    legal[3] = 0                                        # a board without a legal cell
    active = torch.ones(rows, dtype=torch.uint8, device="cuda")
    active[7::5] = 0
    logits = torch.full((rows, C_), float("nan"), device="cuda")
    bad = torch.zeros((), dtype=torch.long, device="cuda")
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    a = mlp_actions(pol.packed(), C_, hidden[0], hidden[1], obs, legal, row_active=active, logits_out=logits,
                    bad_rows=bad, rng=(st, 1234, 9, False))
    with torch.no_grad():
        ref = pol.action_net(pol.policy_net(obs.float()))
    on = active.bool()
    assert bool(torch.isnan(logits[~on]).all())         # skipped boards: not evaluated, action 0
    assert bool((a[~on] == 0).all())
    err = float((logits[on] - ref[on]).abs().max())
    assert err < 2e-5, err
    bad2 = torch.zeros((), dtype=torch.long, device="cuda")
    b = sample_masked(logits.nan_to_num(0.0), legal, None, bad2, row_active=active, rng=(st, 1234, 9, False))
    assert torch.equal(a, b) and int(bad) == int(bad2) == int(((legal.sum(1) == 0) & on).sum()) >= (1 if bool(on[3]) else 0)
    assert bool(legal[on].gather(1, a[on].long().view(-1, 1)).view(-1)[legal[on].sum(1) > 0].all())
    # arg-max play
    det = mlp_actions(pol.packed(), C_, hidden[0], hidden[1], obs, legal, deterministic=True, logits_out=logits)
    want = logits.masked_fill(legal == 0, float("-inf")).argmax(1)
    ok = legal.sum(1) > 0
    assert torch.equal(det[ok].long(), want[ok])
    # the policy object takes the same path, and follows in-place weight updates
    pol.deterministic = True
    assert pol.fusable and torch.equal(pol.choose_actions(obs.view(rows, n, n), legal), det)
    with torch.no_grad():
        pol.action_net.bias.add_(torch.randn(C_, device="cuda") * 5)
    det2 = pol.choose_actions(obs, legal)
    with torch.no_grad():
        want2 = pol.action_net(pol.policy_net(obs.float())).masked_fill(legal == 0, float("-inf")).argmax(1)
    assert float((det2[ok].long() == want2[ok]).double().mean()) > 0.999 and not torch.equal(det2, det)


def test_fused_actor_serves_a_pool_of_networks_in_one_call(hx):
    """every board names its own network through its opponent slot (SelfplayWrapper.py:97-104): the boards are
    grouped on the device and each gets the logits of ITS network; the result is independent of the grouping"""
    from hex_gym_env_amd.vec_selfplay import MlpActorPolicy, mlp_actions, sample_masked
    torch.manual_seed(3)
    n, rows, P = 7, 5000, 6
    C_ = n * n
    pols = [MlpActorPolicy(C_).cuda() for _ in range(P)]
    table = torch.stack([p_.packed() for p_ in pols])
    obs, legal = _random_position(rows, C_, 11)
    of_slot = torch.tensor([0, 3, 1, 1, 5, 2, 4, 0, 3], dtype=torch.int32, device="cuda")   # 9 slots -> 6 networks
    slot = torch.randint(-1, 8, (rows,), device="cuda")
    slot[:4000] = -1                                     # 80 % on the best model, like the reference
    slot = slot[torch.randperm(rows, device="cuda")].contiguous()
    active = (torch.rand(rows, device="cuda") < 0.7).to(torch.uint8)
    logits = torch.full((rows, C_), float("nan"), device="cuda")
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    a = mlp_actions(table, C_, 64, 64, obs, legal, row_active=active, slot=slot, slot_bias=1, policy_of_slot=of_slot,
                    logits_out=logits, rng=(st, 77, 2, True))
    assert int(st[0]) == 1                               # the call advanced the draw counter
    st.zero_()
    which = of_slot[(slot + 1).long()].long()
    with torch.no_grad():
        ref = torch.stack([p_.action_net(p_.policy_net(obs.float())) for p_ in pols])        # [P, rows, C]
    ref = ref[which, torch.arange(rows, device="cuda")]
    on = active.bool()
    assert float((logits[on] - ref[on]).abs().max()) < 2e-5
    assert bool(torch.isnan(logits[~on]).all()) and bool((a[~on] == 0).all())
    b = sample_masked(logits.nan_to_num(0.0), legal, None, None, row_active=active, rng=(st, 77, 2, False))
    assert torch.equal(a, b)
    # one network per call on the boards of that network == the grouped call
    for u in range(P):
        sel = (which == u) & on
        one = mlp_actions(table[u], C_, 64, 64, obs, legal, row_active=sel.to(torch.uint8), rng=(st, 77, 2, False))
        assert torch.equal(one[sel], a[sel])
    # clamped slots, no active board at all, a tiny batch
    wild = slot.clone()
    wild[:10] = 1000
    mlp_actions(table, C_, 64, 64, obs, legal, slot=wild, slot_bias=1, policy_of_slot=of_slot, rng=(st, 1, 0, False))
    none = mlp_actions(table, C_, 64, 64, obs, legal, row_active=torch.zeros_like(active), slot=slot, slot_bias=1,
                       policy_of_slot=of_slot, rng=(st, 1, 0, False))
    assert bool((none == 0).all())
    few = mlp_actions(table, C_, 64, 64, obs[:3], legal[:3], slot=slot[:3].contiguous(), slot_bias=1, policy_of_slot=of_slot,
                      rng=(st, 77, 2, False))
    assert few.shape == (3,)
    with pytest.raises(ValueError):                     # a table that is not in the packed layout of this shape
        mlp_actions(table[:, :-16].contiguous(), C_, 64, 64, obs, legal)


@pytest.mark.parametrize("mode", ["train", "eval", "sample_board"])
def test_vec_selfplay_pool_of_distinct_networks_is_one_launch_per_round(hx, mode):
    """VecSelfPlayEnv with a pool of DISTINCT actor networks: the weights-table path (one policy launch per
    opponent round) plays exactly what the per-policy grouped path plays (arg-max opponents: no random draws
    to line up), and keeps following pool updates -- in the training step, in evaluation mode (the pool is walked in
    order, SelfplayWrapper.py:92-96) and with sampled start positions (sample_board)"""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
    n, E = 5, 2048
    torch.manual_seed(1)
    nets = [MlpActorPolicy(n * n, deterministic=True).cuda() for _ in range(4)]

    def make(fused):
        env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=6, base_model=nets[0], seed=4,
                             sample_board=mode == "sample_board")
        for k in range(6):
            env.set_opponent_model(k, nets[k % 4], float(k))
        if mode == "eval":
            env.set_eval(True)
        if not fused:
            env._pool_table = lambda: None
        return env

    ea, eb = make(True), make(False)
    assert ea._pool_table() is not None and len(ea._pool_table().policies) == 4
    oa, la = ea.reset()
    ob, lb = eb.reset()
    assert torch.equal(oa, ob) and torch.equal(la, lb)
    g = torch.Generator(device="cuda").manual_seed(9)
    ended = 0
    for t in range(40):
        if t == 20:                                      # a pool update in the middle (SelfPlayCallback's cycle)
            fresh = MlpActorPolicy(n * n, deterministic=True).cuda()
            ea.set_opponent_model(2, fresh, 9.0)
            eb.set_opponent_model(2, fresh, 9.0)
        act = torch.multinomial(la.float() + 1e-6, 1, generator=g).view(-1).to(torch.int32)
        ra, rb = ea.step(act), eb.step(act)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), t
        la = ra[3]
        ended += int(ra[2].sum())
    assert ended > E and ea.degenerate_rows() == 0
    assert torch.equal(ea.slot, eb.slot)


class _FakeSb3Policy(torch.nn.Module):
    """the attribute layout of stable_baselines3's ActorCriticPolicy with the default MLP extractor"""

    def __init__(self, cells, hidden=(64, 64)):
        super().__init__()
        self.features_extractor = torch.nn.Flatten()
        self.mlp_extractor = torch.nn.Module()
        dims = [cells] + list(hidden)
        layers = []
        for a, b in zip(dims, dims[1:]):
            layers += [torch.nn.Linear(a, b), torch.nn.Tanh()]
        self.mlp_extractor.policy_net = torch.nn.Sequential(*layers)
        self.mlp_extractor.value_net = torch.nn.Sequential(torch.nn.Linear(cells, 64), torch.nn.Tanh())
        self.action_net = torch.nn.Linear(dims[-1], cells)
        self.value_net = torch.nn.Linear(64, 1)


class _FakeSb3Model:
    def __init__(self, cells, hidden=(64, 64)):
        self.policy = _FakeSb3Policy(cells, hidden)          # on the CPU, like a model loaded with device="cpu"
        self.saved = None

    def predict(self, obs, deterministic=False, action_masks=None):
        raise AssertionError("the fused view must not call predict()")

    def save(self, path):
        self.saved = path


def test_sb3_style_models_run_on_the_fused_kernel_as_live_views(hx):
    """set_opponent_model(model) with an SB3-style model (OpponentPolicy(model), SelfplayWrapper.py:26-38): its
    default-MLP actor is served by hexvec_mlp_actions through a LIVE view of the model's own layers (CPU or GPU),
    never by per-board predict(); other architectures fall back to predict()"""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy, PredictAdapter, as_policy
    n, E = 5, 1024
    torch.manual_seed(5)
    m0, m1 = _FakeSb3Model(n * n), _FakeSb3Model(n * n)
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=4, base_model=m0, seed=2)
    env.set_opponent_model(1, m1, 3.0)
    v0, v1 = env.buffer.policy(0), env.buffer.policy(1)
    assert isinstance(v0, MlpActorPolicy) and v0.policy_net is m0.policy.mlp_extractor.policy_net
    assert v1.action_net is m1.policy.action_net and v1.compute_device.type == "cuda"
    tab = env._pool_table()
    assert tab is not None and len(tab.policies) == 2
    obs, legal = env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(30):
        act = torch.multinomial(legal.float() + 1e-6, 1, generator=g).view(-1).to(torch.int32)
        obs, rew, done, legal, win = env.step(act)
        assert int((win == 3).sum()) == 0
    assert env.degenerate_rows() == 0
    # the view's logits are the model's, and they follow the model as it trains (in-place updates on the CPU)
    o, l = _random_position(300, n * n, 3)
    lg = torch.empty((300, n * n), device="cuda")
    from hex_gym_env_amd.vec_selfplay import mlp_actions
    mlp_actions(v1.packed(), n * n, 64, 64, o, l, deterministic=True, logits_out=lg)
    with torch.no_grad():
        ref = m1.policy.action_net(m1.policy.mlp_extractor.policy_net(o.float().cpu())).cuda()
    assert float((lg - ref).abs().max()) < 2e-5
    with torch.no_grad():
        m1.policy.action_net.bias.add_(1.0)
    mlp_actions(v1.packed(), n * n, 64, 64, o, l, deterministic=True, logits_out=lg)
    assert float((lg - ref - 1.0).abs().max()) < 2e-5
    v1.save_model("somewhere")
    assert m1.saved == "somewhere"
    env.set_opponent_model(3, m1, 4.0)                   # the same live model again: the same view, still two networks
    assert env.buffer.policy(3) is v1 and len(env._pool_table().policies) == 2
    # a different architecture: predict() adapter
    odd = _FakeSb3Model(n * n, hidden=(64, 64, 64))
    assert isinstance(as_policy(odd, "cuda"), PredictAdapter)


# ---- hexvec_attach_pool: setup_opponents and the openings in the tails of the split step's own launches ----
@pytest.mark.parametrize("agent,openings", [(0, "none"), (None, "launch"), (1, "launch")],
                         ids=["agents-black", "mixed-colours", "agents-white"])
def test_attached_pool_equals_the_separate_pick_launches(hx, agent, openings):
    """SelfPlayEnv.setup_opponents (SelfplayWrapper.py:97-104) folded into hexvec_step_agent / hexvec_step_opponent
    (hexvec_attach_pool) picks the same boards with the same draws as hexvec_pick_opponents as launches of its own,
    and publishes the same draw counter: two envs, same seed, sampled (not arg-max) opponents from a pool of two
    networks -- every output of every step, the slot array and the counter are equal.  (The openings stay a launch
    of their own here; inside the step they are drawn differently, see the next test.)"""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
    n, E = 5, 2048
    torch.manual_seed(3)
    nets = [MlpActorPolicy(n * n).cuda() for _ in range(2)]

    def make(fused):
        env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=6, base_model=nets[0], seed=11, agent_player_num=agent)
        env.set_opponent_model(3, nets[1], 0.1)         # a second network in the pool (not the best model)
        env.fuse_pool = fused
        if fused:
            env._opening_policies = lambda: None        # picks and counter in the step's tails, openings as a launch
        return env

    ea, eb = make(True), make(False)
    oa, la = ea.reset()
    ob, lb = eb.reset()
    assert torch.equal(oa, ob)
    g = torch.Generator(device="cuda").manual_seed(5)
    ended = 0
    for t in range(60):
        act = torch.multinomial(la.float() + 1e-6, 1, generator=g).view(-1).to(torch.int32)
        ra, rb = ea.step(act), eb.step(act)
        assert ea._attached is not None and eb._attached is None
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert torch.equal(x, y), (t, k)
        assert torch.equal(ea.slot, eb.slot), t
        assert torch.equal(ea._pick_state, eb._pick_state) and int(ea._pick_state[0]) == t + 1
        assert int(ea.vec.pending.sum()) == 0
        la = ra[3]
        ended += int(ra[2].sum())
    assert ended > 3 * E and len(set(ea.slot.tolist())) == 7          # -1 and all six members were drawn
    assert ea.degenerate_rows() == 0 and eb.degenerate_rows() == 0


@pytest.mark.parametrize("E", [16384, 16384 + 37], ids=["full-wavefronts", "ragged-last-wavefront"])
def test_openings_inside_the_step_follow_each_opponents_opening_distribution(hx, E):
    """with an opening distribution per pool member (hexvec_attach_pool(open_cdf)), ONE hexvec_step_agent + ONE
    hexvec_step_opponent complete a step: a board that ends on the reply gets its new opponent and that opponent's
    opening inside the same launch (SelfplayWrapper.py:75, :79-80).  All agents WHITE: every fresh board shows exactly
    the opening stone; per pool member the empirical opening frequencies match softmax(logits(empty board)) -- a
    sampled network, an arg-max network (always its best cell) and the uniform random policy."""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy, RandomLegalPolicy
    n, C = 4, 16
    torch.manual_seed(8)
    sampled, greedy = MlpActorPolicy(C).cuda(), MlpActorPolicy(C, deterministic=True).cuda()
    with torch.no_grad():
        sampled.action_net.bias.mul_(0).add_(torch.linspace(-1.5, 1.5, C, device="cuda"))      # a clearly non-uniform opening
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=4, base_model=sampled, seed=2, agent_player_num=1,
                         prob_best=0.4)
    env.set_opponent_model(1, greedy, 0.0)
    env.set_opponent_model(2, RandomLegalPolicy(env._gen), 0.0)
    assert env.best_model is sampled and env._pool_table() is None and len(env._groups()) == 3
    env.reset()
    counts = {k: torch.zeros(C, dtype=torch.long, device="cuda") for k in ("sampled", "greedy", "random")}
    kind_of_slot = {-1: "sampled", 0: "sampled", 1: "greedy", 2: "random", 3: "sampled"}
    g = torch.Generator(device="cuda").manual_seed(1)
    legal = env.vec.legal
    launches = []
    for t in range(50):
        act = torch.multinomial(legal.float() + 1e-6, 1, generator=g).view(-1).to(torch.int32)
        obs, rew, done, legal, win = env.step(act)
        launches.append(env.vec.last_kernel())
        assert env._attached is not None and int(env.vec.pending.sum()) == 0
        o = obs.reshape(E, n, n)
        fresh = done != 0
        assert bool(((o[fresh] != 0).sum((1, 2)) == 1).all()) and bool(((o[fresh] == 1).sum((1, 2)) == 1).all())
        # the opponent (BLACK) chose its cell in ITS view; the WHITE agent sees the transposed board
        cell = (o[fresh].transpose(1, 2).reshape(-1, C) == 1).float().argmax(1)
        slots = env.slot[fresh]
        for s, kind in kind_of_slot.items():
            counts[kind] += torch.bincount(cell[slots == s], minlength=C)
    tot = {k: int(v.sum()) for k, v in counts.items()}
    assert min(tot.values()) > 10000, tot
    p = torch.softmax(sampled.opening_logits().reshape(-1).double(), 0)
    f = counts["sampled"].double() / tot["sampled"]
    assert float((f - p).abs().max()) < 4.0 * float((p * (1 - p) / tot["sampled"]).sqrt().max()) + 1e-3
    best = int(greedy.opening_logits().argmax())
    assert int(counts["greedy"][best]) == tot["greedy"]
    f = counts["random"].double() / tot["random"]
    assert float((f - 1.0 / C).abs().max()) < 4.0 * (1.0 / C / tot["random"]) ** 0.5 + 1e-3
    assert env.degenerate_rows() == 0
    # a network whose opening cannot be sampled (a NaN logit) is counted, never silent
    with torch.no_grad():
        sampled.action_net.bias[3] = float("nan")
    before = env.degenerate_rows()
    for t in range(3):
        act = torch.multinomial(legal.float() + 1e-6, 1, generator=g).view(-1).to(torch.int32)
        legal = env.step(act)[3]
    assert env.degenerate_rows() > before


def test_attach_pool_rejects_other_handles(hx):
    import ctypes as C
    L = hx._lib.lib()
    slot = torch.zeros(8, dtype=torch.long, device="cuda")
    ended = torch.zeros(8, dtype=torch.uint8, device="cuda")
    st = torch.zeros(2, dtype=torch.long, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    for kw in (dict(rollout_layout=True), dict(two_sided=True), dict(auto_reset=False)):
        env = hx.HexVecEnv(5, 8, encoding="B", **kw)
        rc = L.hexvec_attach_pool(env._h, p(slot), p(ended), C.c_float(0.8), 4, C.c_uint64(1), p(st), None, None, 1, None)
        assert rc == -4, kw
    env = hx.HexVecEnv(5, 8, encoding="B")
    assert L.hexvec_attach_pool(env._h, p(slot), None, C.c_float(0.8), 4, C.c_uint64(1), p(st), None, None, 1, None) == -1
    assert L.hexvec_attach_pool(env._h, p(slot), p(ended), C.c_float(1.5), 4, C.c_uint64(1), p(st), None, None, 1, None) == -1
    assert L.hexvec_attach_pool(env._h, p(slot), p(ended), C.c_float(0.8), 4, C.c_uint64(1), p(st), None, None, 1, None) == 0
    assert L.hexvec_attach_pool(env._h, None, None, C.c_float(0.0), 1, C.c_uint64(0), None, None, None, 1, None) == 0


def test_raw_pointer_wrappers_validate_their_tensors(hx):
    """sample_masked / mlp_actions hand raw pointers to the library: a mistyped, misplaced or short tensor must raise
    here instead of becoming an out-of-bounds access on the GPU; a network whose fused kernel would not fit the LDS
    (256-256 at 16x16) is served by the torch forward instead of failing"""
    from hex_gym_env_amd.vec_selfplay import MlpActorPolicy, mlp_actions, sample_masked
    rows, C = 64, 25
    legal = torch.ones((rows, C), dtype=torch.uint8, device="cuda")
    logits = torch.zeros((rows, C), device="cuda")
    for bad in (dict(logits=torch.zeros((rows, C + 1), device="cuda")), dict(logits=logits.cpu()),
                dict(out=torch.empty(rows, dtype=torch.int64, device="cuda")),
                dict(out=torch.empty(rows - 1, dtype=torch.int32, device="cuda")),
                dict(row_active=torch.ones(rows + 3, dtype=torch.uint8, device="cuda")),
                dict(bad_rows=torch.zeros((), dtype=torch.int32, device="cuda"))):
        kw = dict(logits=logits)
        kw.update(bad)
        with pytest.raises(ValueError):
            sample_masked(kw.pop("logits"), legal, **kw)
    pol = MlpActorPolicy(C).cuda()
    obs = torch.zeros((rows, 5, 5), dtype=torch.int8, device="cuda")
    w = pol.packed()
    for bad in (dict(out=torch.empty(rows - 1, dtype=torch.int32, device="cuda")),
                dict(slot=torch.zeros(rows - 1, dtype=torch.long, device="cuda")),
                dict(logits_out=torch.empty((rows, C - 1), device="cuda")),
                dict(row_active=torch.ones(rows - 1, dtype=torch.uint8, device="cuda"))):
        with pytest.raises(ValueError):
            mlp_actions(w, C, 64, 64, obs, legal, **bad)
    with pytest.raises(ValueError):
        mlp_actions(w, C, 64, 64, obs[:-1], legal)
    assert hx._lib.lib().hexvec_mlp_lds_bytes(256, 256, 256) > 160 * 1024
    big = MlpActorPolicy(256, hidden=(256, 256), deterministic=True).cuda()
    assert not big.fusable and MlpActorPolicy(121, hidden=(256, 256)).fusable
    obs16 = torch.zeros((8, 16, 16), dtype=torch.int8, device="cuda")
    legal16 = torch.ones((8, 256), dtype=torch.uint8, device="cuda")
    a = big.choose_actions(obs16, legal16)
    assert torch.equal(a.long(), big.logits(obs16, legal16).argmax(1))


def test_attach_pool_on_a_hex_v0_handle_through_the_c_abi(hx):
    """hexvec_attach_pool below the Python wrapper, on an encoding-A self-play handle where the opponent opens every
    episode (FIRST_WHITE): one hexvec_step_agent + one hexvec_step_opponent complete a step (pending all zero), a board
    that ended shows exactly the opening stone of the policy its NEW slot names (two one-hot opening tables), the slot of
    a board changes only when its episode ends, and the published counter counts the steps"""
    import ctypes as C
    n, E, Cc = 5, 4096, 25
    env = hx.HexVecEnv(n, E, encoding="A", first_player=1, seed=6)
    L = hx._lib.lib()
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    slot = torch.full((E,), -1, dtype=torch.long, device="cuda")
    ended = torch.zeros(E, dtype=torch.uint8, device="cuda")
    state = torch.zeros(2, dtype=torch.long, device="cuda")
    bad = torch.zeros((), dtype=torch.long, device="cuda")
    cdf = torch.zeros((2, Cc), device="cuda")
    cdf[0, 7:] = 1.0            # policy 0 always opens at cell 7 (of ITS view), policy 1 at cell 18
    cdf[1, 18:] = 1.0
    pos = torch.tensor([0, 1, 0, 1], dtype=torch.int32, device="cuda")     # slot -1 -> policy 0, 0 -> 1, 1 -> 0, 2 -> 1
    assert L.hexvec_attach_pool(env._h, p(slot), p(ended), C.c_float(0.5), 3, C.c_uint64(99), p(state), p(cdf), p(pos), 2, p(bad)) == 0
    obs, legal = env.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    seen = torch.zeros(2, dtype=torch.long, device="cuda")
    for t in range(40):
        prev_slot = slot.clone()
        act = torch.multinomial(legal.float() + 1e-6, 1, generator=g).view(-1).to(torch.int32)
        o1, l1, pend = env.step_agent(act)
        opp = torch.multinomial(l1.float() + 1e-6, 1, generator=g).view(-1).to(torch.int32)
        # boards that ended on the agent's ply already have their new slot; their opening is the caller's to supply
        # with the replies (the wrapper's opponent round does that): the cell of the policy the slot names
        owed = pend == 2
        opp = torch.where(owed, torch.where(pos[(slot + 1).long()] == 0, 7, 18).to(torch.int32), opp)
        obs, rew, done, legal, win, pend = env.step_opponent(opp)
        assert int(pend.sum()) == 0 and int(state[0]) == t + 1
        fresh = done != 0
        assert bool((slot[~fresh] == prev_slot[~fresh]).all())
        o = obs.reshape(E, Cc)
        assert bool(((o[fresh] != 2).sum(1) == 1).all())            # exactly the opening stone (WHITE = 1) on a fresh board
        # the opponent chose in ITS view (the transposed board): cell c there is cell T(c) here
        cell = (o[fresh].reshape(-1, n, n).transpose(1, 2).reshape(-1, Cc) == 1).float().argmax(1)
        want = torch.where(pos[(slot[fresh] + 1).long()] == 0, 7, 18)
        assert torch.equal(cell, want), t
        seen += torch.bincount(pos[(slot[fresh] + 1).long()].long(), minlength=2)
    assert int(seen.min()) > 100 and int(bad) == 0 and set(slot.tolist()) == {-1, 0, 1, 2}


def test_openings_inside_the_step_single_policy_table_in_registers(hx):
    """one opening distribution for the whole pool (the reference's usual pool: the same live model in every slot): the
    launch holds the table in registers (requested before the ply); frequencies match softmax(logits(empty board))"""
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
    n, E, C = 5, 8192, 25
    torch.manual_seed(21)
    pol = MlpActorPolicy(C).cuda()
    with torch.no_grad():
        pol.action_net.bias.mul_(0).add_(torch.linspace(1.0, -2.0, C, device="cuda"))
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=4, base_model=pol, seed=12, agent_player_num=1)
    env.reset()
    assert len(env._groups()) == 1
    counts = torch.zeros(C, dtype=torch.long, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(2)
    legal = env.vec.legal
    for t in range(60):
        act = torch.multinomial(legal.float() + 1e-6, 1, generator=g).view(-1).to(torch.int32)
        obs, rew, done, legal, win = env.step(act)
        assert env._attached is not None and env._attached[2] is not None and int(env.vec.pending.sum()) == 0
        fresh = done != 0
        o = obs.reshape(E, n, n)[fresh]
        assert bool(((o != 0).sum((1, 2)) == 1).all())
        counts += torch.bincount((o.transpose(1, 2).reshape(-1, C) == 1).float().argmax(1), minlength=C)
    tot = int(counts.sum())
    assert tot > 20000
    p = torch.softmax(pol.opening_logits().reshape(-1).double(), 0)
    f = counts.double() / tot
    assert float((f - p).abs().max()) < 4.0 * float((p * (1 - p) / tot).sqrt().max()) + 1e-3
    assert env.degenerate_rows() == 0


def test_capture_collect_n_steps_equals_n_single_step_replays(hx):
    """capture_collect(agent, n): n collection steps per hipGraph, every step's rows written straight into the rollout
    buffer (step(out=...): the split step's launches write the caller's rows).  Same launches and same draws as n
    replays of capture_collect_step(): from one common state, 8 single-step replays == 2 replays of a 4-step graph, row
    for row (what the agent saw, what it played, reward, done, winner), and the envs end in the same state"""
    import copy
    from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
    n, E, K = 5, 4096, 4
    torch.manual_seed(5)
    agent, opp, opp2 = (MlpActorPolicy(n * n).cuda() for _ in range(3))

    def make():
        env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=5, base_model=opp, seed=31)
        env.set_opponent_model(1, opp2, 0.0)      # two networks in the pool: the table path with per-slot openings
        env.reset()
        return env

    ea, eb = make(), make()
    ag_a, ag_b = agent, copy.deepcopy(agent)
    run1 = ea.capture_collect_step(ag_a)
    runk = eb.capture_collect(ag_b, K)
    # one common state for both: env blob (boards, RNG counters, the handle's step counters), slots, the draw counter,
    # the current observation
    eb.vec.load_state_dict(ea.vec.state_dict())
    eb.slot.copy_(ea.slot)
    eb._pick_state.copy_(ea._pick_state)
    eb.vec.obs.copy_(ea.vec.obs)
    eb.vec.legal.copy_(ea.vec.legal)
    rows = []
    for t in range(2 * K):
        seen_obs, seen_legal = ea.vec.obs.clone(), ea.vec.legal.clone()
        obs, rew, done, legal, win, act = run1()
        rows.append((seen_obs, seen_legal, act.clone(), rew.clone(), done.clone(), win.clone()))
    for rep in range(2):
        buf = runk()
        for t in range(K):
            so, sl, a, r, d, w = rows[rep * K + t]
            assert torch.equal(buf["obs"][t], so) and torch.equal(buf["legal"][t], sl), (rep, t)
            assert torch.equal(buf["actions"][t], a.to(torch.int32)), (rep, t)
            assert torch.equal(buf["reward"][t], r) and torch.equal(buf["done"][t], d) and torch.equal(buf["winner"][t], w), (rep, t)
    assert torch.equal(ea.vec.obs, eb.vec.obs) and torch.equal(ea.vec.legal, eb.vec.legal)
    assert torch.equal(ea.slot, eb.slot) and torch.equal(ea._pick_state, eb._pick_state)
    assert torch.equal(ea.vec.state_dict()["blob"], eb.vec.state_dict()["blob"])
    assert int(sum(r[4].sum() for r in rows)) > E // 4 and runk.captures == 1
    assert ea.degenerate_rows(ag_a) == 0 and eb.degenerate_rows(ag_b) == 0
    with pytest.raises(NotImplementedError):            # evaluation mode keeps the general step: no caller rows there
        eb.set_eval(True)
        eb.step(torch.zeros(E, dtype=torch.int32, device="cuda"), out=dict(reward=torch.empty(E, device="cuda")))
