"""run_actor_only.py -- N launches of the fused actor kernel alone (7x7 / 32 768 by default): the program for a rocprofv3
counter pass on that kernel (tools/pmc_actor.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hex_gym_env_amd.vec_selfplay import MlpActorPolicy, mlp_actions

dev = torch.device("cuda", 0)
n, E, L = int(os.environ.get("N", 7)), int(os.environ.get("E", 32768)), int(os.environ.get("LAUNCHES", 200))
C = n * n
torch.manual_seed(0)
pol = MlpActorPolicy(C).to(dev)
obs = (torch.randint(0, 3, (E, C), device=dev) - 1).to(torch.int8)
legal = (obs == 0).to(torch.uint8)
st = torch.zeros(2, dtype=torch.int64, device=dev)
out = torch.empty(E, dtype=torch.int32, device=dev)
w = pol.packed()
for _ in range(L):
    mlp_actions(w, C, 64, 64, obs, legal, rng=(st, 1, 0, True), out=out)
torch.cuda.synchronize()
print("done")
