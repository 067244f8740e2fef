#!/usr/bin/env python3
"""Two context measurements for DESIGN.md (run on the GPU box):
 (1) streaming-write ceiling: torch fill / copy of buffers as large as the rollout outputs;
 (2) BASELINE config 3 (7x7, 32 768 envs, SelfplayWrapper + MLP policy): share of the env in a
     PPO-style collection step (policy forward + masked sampling + env step with MLP opponents)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda", 0)


def ev_time(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


# (1)
nbytes = 2 * 1024 ** 3
a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
t = ev_time(lambda: a.fill_(7), 20)
print("fill_  %.2f GB in %.3f ms -> %.2f TB/s written" % (nbytes / 1e9, t * 1e3, nbytes / t / 1e12))
a32, b32 = a.view(torch.int32), b.view(torch.int32)
t = ev_time(lambda: b32.copy_(a32), 20)
print("copy_  %.2f GB in %.3f ms -> %.2f TB/s read+written" % (nbytes / 1e9, t * 1e3, 2 * nbytes / t / 1e12))

# (2)
from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
n, E = 7, 32768
torch.manual_seed(0)
agent = MlpActorPolicy(n * n).to(dev)
opp = MlpActorPolicy(n * n).to(dev)
env = VecSelfPlayEnv(board_size=n, num_envs=E, buffer_size=20, base_model=opp, seed=1)
obs, legal = env.reset()
state = {"obs": obs, "legal": legal}


def policy_only():
    state["a"] = agent.choose_actions(state["obs"], state["legal"])


def env_only():
    o, r, d, l, w = env.step(state["a"])
    state["obs"], state["legal"] = o, l


def both():
    policy_only()
    env_only()


policy_only()
tp = ev_time(policy_only, 50)
tb = ev_time(both, 50)
print("config 3 (7x7, %d envs): agent policy forward+sample %.1f us; full collection step %.1f us "
      "(env incl. ~2 opponent-policy calls %.1f us) -> %.2f M agent steps/s" %
      (E, tp * 1e6, tb * 1e6, (tb - tp) * 1e6, E / tb / 1e6))
run = env.capture_collect_step(agent)
bad_rows = torch.zeros((), dtype=torch.long, device=dev)
illegal = torch.zeros((), dtype=torch.long, device=dev)
for _ in range(100):                      # diagnostics on the replayed graph, no device-side asserts anywhere
    obs_, rew_, done_, legal_, win_, act_ = run()
    bad_rows += (legal_.sum(1) == 0).sum()
    illegal += (win_ == 3).sum()
print("  graph replay diagnostics over 100 steps: rows without a legal cell %d, illegal-move endings %d"
      % (int(bad_rows), int(illegal)))
tg = ev_time(run, 200)
print("  the same collection step captured as one hipGraph: %.1f us -> %.2f M agent steps/s" % (tg * 1e6, E / tg / 1e6))
raw = env.vec
a32 = state["a"].to(torch.int32)
tr = ev_time(lambda: raw.step(a32), 200)
print("  same boards, fused hexvec_step with the built-in uniform opponent: %.1f us per step" % (tr * 1e6))
