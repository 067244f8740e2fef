"""one-off deep differential soak on the GPU box: thousands of envs x thousands of steps against the oracle,
both throughput layouts, both encodings, RNG actions with external (partly illegal) ones mixed in, rollouts and
host-stepped calls interleaved on the same handle"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hex_gym_env_amd as hx
from oracle import hex_oracle as ho

def flags(enc, agent, first):
    if enc == "B":
        return ho.F_ENC_B | (ho.F_AGENT_RANDOM if agent is None else (ho.F_AGENT_WHITE if agent else 0))
    return ho.F_FIRST_WHITE if first else 0

t0 = time.time()
for n, E, K in ((11, 4096, 1500), (7, 2048, 1500), (13, 1024, 1200), (5, 2048, 1000), (14, 512, 800)):
    for lay in (False, True):
        for enc, agent, first in (("B", None, 0), ("A", None, 1)):
            g = hx.HexVecEnv(n, E, encoding=enc, agent_player_num=agent, first_player=first, seed=99, rollout_layout=lay)
            o = ho.OracleVec(n, E, seed=99, flags=flags(enc, agent, first))
            o.set_threads(os.cpu_count())
            rng = np.random.default_rng(n + int(lay))
            k = 0
            while k < K:
                if k % 50 == 10:                      # a device-resident rollout in between
                    T = 17
                    out = g.rollout_selfplay(T)
                    for t in range(T):
                        oo, ol, orw, od, ow = o.step()
                        assert np.array_equal(out["obs"][t].cpu().numpy(), oo), (n, lay, enc, k, t)
                        assert np.array_equal(out["done"][t].cpu().numpy(), od), (n, lay, enc, k, t)
                    k += T
                    continue
                a = rng.integers(0, n * n, size=E).astype(np.int32) if k % 9 == 4 else None
                obs, rew, done, legal, win = g.step(a)
                oo, ol, orw, od, ow = o.step(a)
                assert np.array_equal(obs.cpu().numpy().reshape(E, -1), oo), (n, lay, enc, k)
                assert np.array_equal(legal.cpu().numpy(), ol), (n, lay, enc, k)
                assert np.array_equal(rew.cpu().numpy(), orw) and np.array_equal(done.cpu().numpy(), od), (n, lay, enc, k)
                assert np.array_equal(win.cpu().numpy(), ow), (n, lay, enc, k)
                k += 1
            assert np.array_equal(g.episode_stats().cpu().numpy(), o.episode_stats())
            print("ok %2dx%-2d E=%5d layout=%-9s enc %s: %d steps, %d episodes (%.0f s)" % (
                n, n, E, "rollout" if lay else "bitboard", enc, K, o.episode_stats()[2], time.time() - t0), flush=True)

# two-sided handles: the W2 rollout (helper wavefront, scalars of a ply travelling with the next emission) over many
# launches of odd lengths, interleaved with host-stepped plies, both layouts, both encodings
for n, E in ((11, 4096), (9, 2048), (13, 1024), (12, 1024 + 37), (5, 2048)):
    for lay in (False, True):
        for enc in ("B", "A"):
            g = hx.HexVecEnv(n, E, encoding=enc, two_sided=True, seed=77, rollout_layout=lay)
            o = ho.OracleVec(n, E, seed=77, flags=(ho.F_ENC_B if enc == "B" else 0) | ho.F_TWO_SIDED)
            o.set_threads(os.cpu_count())
            for launch in range(14):
                T = (1, 2, 37, 64, 5, 100, 3, 64, 64, 7, 1, 1, 150, 33)[launch]
                out = g.rollout_random(T)
                oobs, olegal, oact, orew, odone = o.rollout(T)
                for k, ref in (("obs", oobs), ("legal", olegal), ("actions", oact), ("reward", orew), ("done", odone)):
                    assert np.array_equal(out[k].cpu().numpy(), ref), (n, lay, enc, launch, k)
                if launch % 4 == 1:
                    obs, rew, done, legal, win = g.step()
                    oo, ol, orw, od, ow = o.step()
                    assert np.array_equal(obs.cpu().numpy().reshape(E, -1), oo) and np.array_equal(done.cpu().numpy(), od), (n, lay, enc, launch)
            assert np.array_equal(g.episode_stats().cpu().numpy(), o.episode_stats())
            print("ok two-sided %2dx%-2d E=%5d layout=%-9s enc %s: %d episodes (%.0f s)" % (
                n, n, E, "rollout" if lay else "bitboard", enc, o.episode_stats()[2], time.time() - t0), flush=True)
