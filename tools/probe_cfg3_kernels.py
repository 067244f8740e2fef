import sys; sys.path.insert(0, ".")
import torch
from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
n, E = 7, 32768
torch.manual_seed(0)
dev = "cuda"
agent = MlpActorPolicy(n * n).to(dev); opp = MlpActorPolicy(n * n).to(dev)
env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=20, base_model=opp, seed=1)
env.reset()
for _ in range(5):
    a = agent.choose_actions(env.vec.obs, env.vec.legal); env.step(a)
torch.cuda.synchronize()
print("MARK")
for _ in range(20):
    a = agent.choose_actions(env.vec.obs, env.vec.legal); env.step(a)
torch.cuda.synchronize()
