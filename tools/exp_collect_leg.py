import sys; sys.path.insert(0, "/root/repo")
import torch, time
from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n3, E3 = 7, 32768
torch.manual_seed(0)
agent, opp = MlpActorPolicy(n3 * n3).to(dev), MlpActorPolicy(n3 * n3).to(dev)
env3 = VecSelfPlayEnv(board_size=n3, num_envs=E3, buffer_size=20, base_model=opp, seed=1, device=dev)
print("table", env3._pool_table() is not None, env3.device, opp.action_net.weight.device)
env3.reset()
run = env3.capture_collect_step(agent)
for _ in range(200):
    run()
torch.cuda.synchronize()
print("captures", run.captures)
g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g0.record()
for _ in range(500):
    run()
g1.record(); torch.cuda.synchronize()
print("us", 1e3 * g0.elapsed_time(g1) / 500, "captures", run.captures)
t0=time.perf_counter()
for _ in range(500):
    run()
torch.cuda.synchronize()
print("wall us", (time.perf_counter()-t0)/500*1e6)
