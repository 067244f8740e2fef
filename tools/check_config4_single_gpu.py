import sys, time; sys.path.insert(0,'.')
import torch, hex_gym_env_amd as hx
n, E, T = 11, 524288, 8
for two in (True, False):
    env = hx.HexVecEnv(n, E, encoding="B", two_sided=two, seed=2)
    fn = env.rollout_random if two else env.rollout_selfplay
    out = fn(T)
    torch.cuda.synchronize()
    obs, legal, done = out["obs"], out["legal"], out["done"]
    assert torch.equal(legal, (obs == 0).to(torch.uint8))
    t0 = time.perf_counter(); out = fn(T, out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = env.episode_stats().cpu().numpy()
    print("two_sided" if two else "selfplay", "E=%d T=%d ok: %.2f G units/s, episodes %d" % (E, T, E*T/dt/1e9, st[2]))
    # sharding: last eighth as its own shard
    sh = hx.HexVecEnv(n, E // 8, encoding="B", two_sided=two, seed=2, env_offset=7 * E // 8)
    o2 = (sh.rollout_random if two else sh.rollout_selfplay)(T)
    env2 = hx.HexVecEnv(n, E, encoding="B", two_sided=two, seed=2)
    o1 = (env2.rollout_random if two else env2.rollout_selfplay)(T)
    assert torch.equal(o1["obs"][:, 7 * E // 8:], o2["obs"]) and torch.equal(o1["actions"][:, 7 * E // 8:], o2["actions"])
    print("  shard 7/8 identical to the monolithic run")
    del env, env2, sh, out, o1, o2
