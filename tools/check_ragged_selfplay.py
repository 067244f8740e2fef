"""sanity run on the GPU box: VecSelfPlayEnv with a pool of actor networks at ragged env counts (1, 37, 65, 130, 1000)
and board sizes 5..19, eager and as a captured collection step: no illegal ply, no degenerate sampling row"""
import sys; sys.path.insert(0, "/root/repo")
import torch
from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
torch.manual_seed(0)
for n, E in ((5, 1), (7, 37), (11, 1000), (13, 65), (19, 130)):
    nets = [MlpActorPolicy(n * n).cuda() for _ in range(3)]
    env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=5, base_model=nets[0], seed=3)
    env.set_opponent_model(1, nets[1], 1.0); env.set_opponent_model(4, nets[2], 2.0)
    agent = MlpActorPolicy(n * n).cuda()
    obs, legal = env.reset()
    ended = 0
    for t in range(3 * n * n):
        a = agent.choose_actions(obs, legal)
        assert bool(legal.gather(1, a.long().view(-1, 1)).all()), (n, E, t)
        obs, rew, done, legal, win = env.step(a)
        assert int((win == 3).sum()) == 0 and bool((legal.sum(1) > 0).all())
        ended += int(done.sum())
    run = env.capture_collect_step(agent)
    for _ in range(20):
        o, r, d, l, w, act = run()
        assert int((w == 3).sum()) == 0
    assert env.degenerate_rows(agent) == 0
    print("ok", n, E, "episodes", ended, "networks", len(env._pool_table().policies))
