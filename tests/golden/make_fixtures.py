#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the live reference.

Runs ONLY in the build container: it needs /root/reference (read-only) and is
never imported by the tests.  The reference files are imported unmodified;
`gymnasium` and `pygame` are absent from this image, so empty stand-ins are
put in sys.modules first (HexGame.py:1-8,
HexSingleGame.py:1-8, SelfplayWrapper.py:1-8 import them at module top,
interactive/gui.py:195 evaluates pygame.Color at import).

The fixtures are DATA ONLY: seeded inputs (action sequences) and the outputs
the reference produced for them.  Everything is small integer arrays written
with numpy.savez_compressed; render() text is captured stdout.

    python tests/golden/make_fixtures.py          # rewrites tests/golden/*.npz, *.txt
"""
import contextlib
import io
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    gym = types.ModuleType("gymnasium")
    gym.Env = type("Env", (), {})
    sp = types.ModuleType("gymnasium.spaces")
    sp.Box = lambda low, high, shape, dtype: types.SimpleNamespace(
        low=low, high=high, shape=shape, dtype=dtype)
    sp.Discrete = lambda n: types.SimpleNamespace(n=n)
    envs = types.ModuleType("gymnasium.envs")
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = lambda **k: None
    pg = types.ModuleType("pygame")
    pg.Color = lambda *a: None
    gym.spaces, gym.envs, envs.registration = sp, envs, reg
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": sp,
                        "gymnasium.envs": envs,
                        "gymnasium.envs.registration": reg, "pygame": pg})


def _ret_code(w):
    """make_move return value -> int8: -1 None, 0/1 winner, 3 illegal."""
    return -1 if w is None else int(w)


def _pack(mask):
    return np.packbits(np.asarray(mask, dtype=np.uint8))


# --------------------------------------------------------------------------
# engine A  (minihex/HexGame.py:16-142)
# --------------------------------------------------------------------------
def gen_engine_a(HexGameA, playerA, sizes, rng):
    out = {}
    for n in sizes:
        n_games = 24 if n <= 5 else (10 if n <= 7 else 6)
        for g in range(n_games):
            dirty = g % 2 == 1
            first = playerA.BLACK if g % 4 < 2 else playerA.WHITE
            board = playerA.EMPTY * np.ones((n, n))
            sim = HexGameA(first, board, playerA.BLACK)
            acts, rets, dones, curs, boards, masks = [], [], [], [], [], []
            extra = 0
            while True:
                empties = [a for a in range(n * n) if sim.is_valid_move(a)]
                occupied = [a for a in range(n * n) if not sim.is_valid_move(a)]
                if not empties:
                    break
                if dirty and occupied and rng.random() < 0.15:
                    a = rng.choice(occupied)          # illegal ply (q1)
                else:
                    a = rng.choice(empties)
                w = sim.make_move(a)
                acts.append(a)
                rets.append(_ret_code(w))
                dones.append(bool(sim.done))
                curs.append(int(sim.current_player_num))
                boards.append(sim.board.astype(np.int8).ravel().copy())
                masks.append(_pack([sim.is_valid_move(i) for i in range(n * n)]))
                if sim.done:
                    if not dirty or extra >= 4:      # q2: plies after the win
                        break
                    extra += 1
            k = f"N{n}_g{g}_"
            out[k + "first"] = np.int8(int(first))
            out[k + "actions"] = np.array(acts, dtype=np.int16)
            out[k + "ret"] = np.array(rets, dtype=np.int8)
            out[k + "done"] = np.array(dones, dtype=np.uint8)
            out[k + "cur_after"] = np.array(curs, dtype=np.int8)
            out[k + "board"] = np.stack(boards)
            out[k + "mask"] = np.stack(masks)
            out[k + "regions"] = sim.regions.astype(np.uint8)
        out[f"N{n}_ngames"] = np.int32(n_games)
    out["sizes"] = np.array(sizes, dtype=np.int32)
    return out


# --------------------------------------------------------------------------
# env B  (minihex/HexSingleGame.py:156-271) -- single-ply env, both colours
# --------------------------------------------------------------------------
def gen_env_b(HexEnvB, sizes, rng):
    out = {}
    for n in sizes:
        n_games = 24 if n <= 5 else (10 if n <= 7 else 6)
        for g in range(n_games):
            dirty = g % 2 == 1
            env = HexEnvB(board_size=n)
            obs0 = env.reset().astype(np.int8).ravel().copy()   # reset() returns a live reference
            acts, rets, r0s, dones, curs, obs, masks = [], [], [], [], [], [], []
            extra = 0
            steps = 0
            while steps < 3 * n * n:
                legal = env.legal_actions()
                empties = [a for a in range(n * n) if legal[a]]
                occupied = [a for a in range(n * n) if not legal[a]]
                if not empties:
                    break
                if dirty and occupied and rng.random() < 0.15:
                    a = rng.choice(occupied)
                else:
                    a = rng.choice(empties)
                o, rew, done, _ = env.step(a)
                steps += 1
                acts.append(a)
                rets.append(_ret_code(env.winner))
                r0s.append(int(rew[0]))
                assert rew[1] == -rew[0]
                dones.append(bool(done))
                curs.append(int(env.simulator.current_player_num))
                obs.append(o.astype(np.int8).ravel().copy())
                masks.append(_pack(env.legal_actions()))
                if done:
                    if not dirty or extra >= 4:
                        break
                    extra += 1
            k = f"N{n}_g{g}_"
            out[k + "reset_obs"] = obs0
            out[k + "actions"] = np.array(acts, dtype=np.int16)
            out[k + "ret"] = np.array(rets, dtype=np.int8)
            out[k + "r_black"] = np.array(r0s, dtype=np.int8)
            out[k + "done"] = np.array(dones, dtype=np.uint8)
            out[k + "engine_cur_after"] = np.array(curs, dtype=np.int8)
            out[k + "obs"] = np.stack(obs)
            out[k + "mask"] = np.stack(masks)
            out[k + "regions"] = env.simulator.regions.astype(np.uint8)
        out[f"N{n}_ngames"] = np.int32(n_games)
    out["sizes"] = np.array(sizes, dtype=np.int32)
    return out


# --------------------------------------------------------------------------
# SelfPlayEnv  (minihex/SelfplayWrapper.py:37-208), BaseRandomPolicy opponents
# --------------------------------------------------------------------------
def gen_selfplay(selfplay_wrapper, HexEnvB, BaseRandomPolicy, sizes, rng):
    out = {}
    rec = []
    orig = BaseRandomPolicy.choose_action

    def recording(self, board, action_mask=None):
        a = orig(self, board, action_mask)
        rec.append(int(a))
        return a

    BaseRandomPolicy.choose_action = recording
    try:
        for n in sizes:
            n_eps = 16 if n <= 5 else 8
            for e in range(n_eps):
                agent = e % 2
                illegal_end = e % 8 >= 6        # agent ends by an occupied cell
                random.seed(1000 * n + e)       # python RNG used by the wrapper
                env = selfplay_wrapper(HexEnvB)(board_size=n, buffer_size=4,
                                                scores=np.zeros(4),
                                                agent_player_num=agent)
                rec.clear()
                obs0, info = env.reset()
                obs0 = obs0.astype(np.int8).ravel().copy()
                reset_opp = rec[0] if rec else -1
                assert len(rec) == (1 if agent == 1 else 0)
                acts, opps, obs, rews, dones, masks = [], [], [], [], [], []
                mask0 = _pack(env.legal_actions())
                t = 0
                while True:
                    legal = env.legal_actions()
                    empties = [a for a in range(n * n) if legal[a]]
                    occupied = [a for a in range(n * n) if not legal[a]]
                    if illegal_end and occupied and t >= 2:
                        a = rng.choice(occupied)
                    else:
                        a = rng.choice(empties)
                    rec.clear()
                    o, r, done, trunc, inf = env.step(a)
                    assert trunc is False and inf == {}
                    acts.append(a)
                    opps.append(rec[0] if rec else -1)
                    obs.append(o.astype(np.int8).ravel().copy())
                    rews.append(int(r))
                    dones.append(bool(done))
                    masks.append(_pack(env.legal_actions()))
                    t += 1
                    if done:
                        break
                k = f"N{n}_e{e}_"
                out[k + "agent"] = np.int8(agent)
                out[k + "reset_opp_action"] = np.int16(reset_opp)
                out[k + "reset_obs"] = obs0
                out[k + "reset_mask"] = mask0
                out[k + "actions"] = np.array(acts, dtype=np.int16)
                out[k + "opp_actions"] = np.array(opps, dtype=np.int16)
                out[k + "obs"] = np.stack(obs)
                out[k + "reward"] = np.array(rews, dtype=np.int8)
                out[k + "done"] = np.array(dones, dtype=np.uint8)
                out[k + "mask"] = np.stack(masks)
            out[f"N{n}_neps"] = np.int32(n_eps)
    finally:
        BaseRandomPolicy.choose_action = orig
    out["sizes"] = np.array(sizes, dtype=np.int32)
    return out


# --------------------------------------------------------------------------
# env A "hex-v0"  (minihex/HexGame.py:145-359), recorded random opponent
# --------------------------------------------------------------------------
def gen_env_a(HexEnvA, playerA, random_policy, sizes, rng):
    out = {}
    for n in sizes:
        n_eps = 12
        for e in range(n_eps):
            first = playerA.BLACK if e % 2 == 0 else playerA.WHITE
            illegal_end = e % 6 >= 4
            rec = []

            def opp(board):
                a = random_policy(board)
                rec.append(int(a))      # action in the opponent's (inverted) view
                return a

            random.seed(7000 * n + e)
            env = HexEnvA(opponent_policy=opp, player_color=playerA.BLACK,
                          current_player_num=first, board_size=n)
            obs0, info = env.reset()
            obs0 = obs0.astype(np.int8).ravel().copy()
            reset_opp = rec[0] if rec else -1
            reset_last = -1 if info["last_move_opponent"] is None else int(info["last_move_opponent"])
            acts, opps, lasts, obs, rews, dones, wins, masks = [], [], [], [], [], [], [], []
            mask0 = _pack(env.get_action_mask())
            t = 0
            while True:
                legal = env.get_action_mask()
                empties = [a for a in range(n * n) if legal[a]]
                occupied = [a for a in range(n * n) if not legal[a]]
                if illegal_end and occupied and t >= 1:
                    a = rng.choice(occupied)
                else:
                    a = rng.choice(empties)
                rec.clear()
                o, r, done, trunc, inf = env.step(a)
                assert trunc is False
                acts.append(a)
                opps.append(rec[0] if rec else -1)
                lasts.append(-1 if inf["last_move_opponent"] is None else int(inf["last_move_opponent"]))
                obs.append(o.astype(np.int8).ravel().copy())
                rews.append(int(r))
                dones.append(bool(done))
                wins.append(_ret_code(inf["winner"]))
                masks.append(_pack(env.get_action_mask()))
                t += 1
                if done:
                    break
            k = f"N{n}_e{e}_"
            out[k + "first"] = np.int8(int(first))
            out[k + "reset_opp_action"] = np.int16(reset_opp)
            out[k + "reset_last_move_opponent"] = np.int16(reset_last)
            out[k + "reset_obs"] = obs0
            out[k + "reset_mask"] = mask0
            out[k + "actions"] = np.array(acts, dtype=np.int16)
            out[k + "opp_actions"] = np.array(opps, dtype=np.int16)
            out[k + "last_move_opponent"] = np.array(lasts, dtype=np.int16)
            out[k + "obs"] = np.stack(obs)
            out[k + "reward"] = np.array(rews, dtype=np.int16)
            out[k + "done"] = np.array(dones, dtype=np.uint8)
            out[k + "winner"] = np.array(wins, dtype=np.int8)
            out[k + "mask"] = np.stack(masks)
        out[f"N{n}_neps"] = np.int32(n_eps)
    out["sizes"] = np.array(sizes, dtype=np.int32)
    return out


# --------------------------------------------------------------------------
# pre-placed boards through the constructor replay (HexGame.py:53-61,
# HexSingleGame.py:57-65): board -> regions
# --------------------------------------------------------------------------
def gen_preplaced(HexGameA, playerA, HexGameB, sizes, rng):
    out = {}
    for n in sizes:
        for g in range(6):
            fill = rng.random() * 0.8
            cells = np.array([rng.random() < fill for _ in range(n * n)])
            colour = np.array([rng.randrange(2) for _ in range(n * n)])
            ba = np.where(cells, colour, 2).astype(float).reshape(n, n)
            bb = np.where(cells, np.where(colour == 0, -1, 1), 0).astype(float).reshape(n, n)
            ga = HexGameA(playerA.BLACK, ba.copy(), playerA.BLACK)
            gb = HexGameB(0, bb.copy())
            k = f"N{n}_g{g}_"
            out[k + "boardA"] = ba.astype(np.int8)
            out[k + "regionsA"] = ga.regions.astype(np.uint8)
            out[k + "counterA"] = ga.region_counter.astype(np.int32)
            out[k + "boardB"] = bb.astype(np.int8)
            out[k + "regionsB"] = gb.regions.astype(np.uint8)
            out[k + "counterB"] = gb.region_counter.astype(np.int32)
    out["sizes"] = np.array(sizes, dtype=np.int32)
    return out


def gen_sample_board(HexEnvB, sizes):
    """HexEnv(sample_board=True): HexSingleGame.py:174-177, :217-222, :300-331 under np.random seeds"""
    out = {}
    for n in sizes:
        for k in range(6):
            np.random.seed(500 + 10 * n + k)
            env = HexEnvB(board_size=n, sample_board=True)
            out[f"N{n}_s{k}_init_board"] = env.initial_board.astype(np.int8)
            b = env.reset()
            out[f"N{n}_s{k}_reset_board"] = b.astype(np.int8).copy()
            out[f"N{n}_s{k}_regions"] = env.simulator.regions.astype(np.uint8)
            b2 = env.reset()
            out[f"N{n}_s{k}_reset2_board"] = b2.astype(np.int8).copy()
    # distribution of random_board over many draws (for the device-side sampler)
    for n in sizes:
        np.random.seed(9000 + n)
        env = HexEnvB(board_size=n)
        cnt, rows, cols = [], [], []
        for _ in range(4000):
            b = env.random_board(np.zeros((n, n)))
            assert (b == -1).sum() == (b == 1).sum()
            cnt.append(int((b != 0).sum()))
            rows.append(int((b != 0).any(1).sum()))
            cols.append(int((b != 0).any(0).sum()))
        out[f"N{n}_stats_stones_hist"] = np.bincount(cnt, minlength=n * n + 1).astype(np.int32)
        out[f"N{n}_stats_rows_hist"] = np.bincount(rows, minlength=n + 1).astype(np.int32)
        out[f"N{n}_stats_cols_hist"] = np.bincount(cols, minlength=n + 1).astype(np.int32)
    out["sizes"] = np.array(sizes, dtype=np.int32)
    return out


def export_actor_weights(zip_path, out_path):
    """actor weights (data) of a shipped SB3 checkpoint -> npz; torch.load(weights_only=True)"""
    import io
    import zipfile
    import torch
    with zipfile.ZipFile(zip_path) as z:
        sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
    keep = {k: v.numpy() for k, v in sd.items()
            if k.startswith("mlp_extractor.policy_net.") or k.startswith("action_net.")}
    np.savez_compressed(out_path, **keep)


def gen_config1(HexGameA, playerA, random_policy, n=5, games=1000):
    """BASELINE config 1: 5x5, reference CPU HexGame, python-`random` policy (seed 0) for both sides;
    per game the action sequence and the make_move return codes (compact)"""
    random.seed(0)
    T = n * n
    acts = np.full((games, T), -1, np.int8)
    rets = np.full((games, T), -2, np.int8)
    for g in range(games):
        sim = HexGameA(playerA.BLACK, playerA.EMPTY * np.ones((n, n)), playerA.BLACK)
        t = 0
        while not sim.done:
            a = random_policy(sim.board)
            w = sim.make_move(a)
            acts[g, t], rets[g, t] = a, _ret_code(w)
            t += 1
    return {"n": np.int32(n), "actions": acts, "ret": rets}


def gen_render(HexEnvA, playerA, HexEnvB, random_policy):
    texts = {}
    for n in (3, 5):
        random.seed(42 + n)
        env = HexEnvA(opponent_policy=random_policy, player_color=playerA.BLACK,
                      board_size=n)
        env.reset()
        env.step(0)
        env.step(n * n - 1 if env.simulator.is_valid_move(n * n - 1) else n)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            env.render()
        texts[f"render_A_N{n}"] = (env.simulator.board.astype(np.int8), buf.getvalue())
        envb = HexEnvB(board_size=n)
        envb.reset()
        envb.step(1)
        envb.step(n + 1)
        envb.step(2)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            envb.render()
        texts[f"render_B_N{n}"] = (envb.simulator.board.astype(np.int8), buf.getvalue())
    return texts


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures are generated in the build container only")
    _install_stubs()
    sys.path.insert(0, REF)
    from minihex import random_policy
    from minihex.HexGame import HexGame as HexGameA, HexEnv as HexEnvA, player as playerA
    from minihex.HexSingleGame import HexGame as HexGameB, HexEnv as HexEnvB
    from minihex.SelfplayWrapper import selfplay_wrapper, BaseRandomPolicy

    rng = random.Random(20241020)
    np.random.seed(20241020)
    sizes = [2, 3, 4, 5, 7, 11, 13]

    np.savez_compressed(os.path.join(OUT, "engineA.npz"),
                        **gen_engine_a(HexGameA, playerA, sizes, rng))
    np.savez_compressed(os.path.join(OUT, "envB.npz"),
                        **gen_env_b(HexEnvB, sizes, rng))
    np.savez_compressed(os.path.join(OUT, "selfplay.npz"),
                        **gen_selfplay(selfplay_wrapper, HexEnvB, BaseRandomPolicy,
                                       [3, 5, 7, 11], rng))
    np.savez_compressed(os.path.join(OUT, "envA.npz"),
                        **gen_env_a(HexEnvA, playerA, random_policy, [3, 5, 7], rng))
    np.savez_compressed(os.path.join(OUT, "preplaced.npz"),
                        **gen_preplaced(HexGameA, playerA, HexGameB, [3, 5, 7, 11], rng))
    np.savez_compressed(os.path.join(OUT, "sample_board.npz"), **gen_sample_board(HexEnvB, [5, 7, 11]))
    np.savez_compressed(os.path.join(OUT, "config1_5x5.npz"), **gen_config1(HexGameA, playerA, random_policy))
    export_actor_weights(os.path.join(REF, "models", "5x5_MLP-default_lr-0.0003_31"),
                         os.path.join(OUT, "sb3_5x5_actor.npz"))
    for name, (board, text) in gen_render(HexEnvA, playerA, HexEnvB, random_policy).items():
        np.save(os.path.join(OUT, name + "_board.npy"), board)
        with open(os.path.join(OUT, name + ".txt"), "w") as f:
            f.write(text)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
