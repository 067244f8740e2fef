#!/usr/bin/env python3
"""Policy side of a self-play collection step (run on the GPU box): hexvec_mlp_actions (forward + mask + sample in
one launch) beside the torch fp32 spelling (3 GEMMs + 2 tanh + casts + hexvec_sample_masked), for one network and
for a pool of distinct networks (the torch spelling evaluates every distinct member on the whole batch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hex_gym_env_amd.vec_selfplay import MlpActorPolicy, mlp_actions, sample_masked

dev = torch.device("cuda", 0)


def ev(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def graphed(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.synchronize()
    return g.replay


for n, E in ((7, 32768), (11, 65536), (13, 16384), (5, 4096)):
    C = n * n
    torch.manual_seed(0)
    pol = MlpActorPolicy(C).to(dev)
    obs = (torch.randint(0, 3, (E, C), device=dev) - 1).to(torch.int8)
    legal = (obs == 0).to(torch.uint8)
    st = torch.zeros(2, dtype=torch.int64, device=dev)
    out = torch.empty(E, dtype=torch.int32, device=dev)
    w = pol.packed()

    def fused():
        mlp_actions(w, C, 64, 64, obs, legal, rng=(st, 1, 0, True), out=out)

    def spelled():
        with torch.no_grad():
            z = pol.action_net(pol.policy_net(obs.float()))
        sample_masked(z, legal, None, None, out=out, rng=(st, 1, 0, True))

    flops = 2.0 * E * (C * 64 + 64 * 64 + 64 * C)
    tf, ts = ev(fused), ev(spelled)
    gf, gs = ev(graphed(fused)), ev(graphed(spelled))
    print("%2dx%-2d %6d boards, one network : fused %6.1f us (graph %6.1f us, %.1f TFLOP/s fp32) | torch %6.1f us (graph %6.1f us)"
          % (n, n, E, tf, gf, flops / gf / 1e6, ts, gs), flush=True)
    # a pool: best model on 80 % of the boards, 20 members on the rest
    P = 21
    pols = [pol] + [MlpActorPolicy(C).to(dev) for _ in range(P - 1)]
    table = torch.stack([p.packed() for p in pols])
    slot = torch.where(torch.rand(E, device=dev) < 0.8, torch.full((E,), -1, device=dev), torch.randint(0, P - 1, (E,), device=dev))
    scratch = None

    def fused_pool():
        mlp_actions(table, C, 64, 64, obs, legal, slot=slot, slot_bias=1, rng=(st, 1, 0, True), out=out)

    def spelled_pool():
        with torch.no_grad():
            for k, p in enumerate(pols):
                z = p.action_net(p.policy_net(obs.float()))
                a = sample_masked(z, legal, None, None, rng=(st, 1, k, False))
                torch.where(slot == k - 1, a, out, out=out)

    tp, tsp = ev(fused_pool, 100), ev(spelled_pool, 20)
    gp = ev(graphed(fused_pool), 100)
    print("      %6d boards, 21 networks : fused %6.1f us (graph %6.1f us) | torch, every member on the whole batch %8.1f us"
          % (E, tp, gp, tsp), flush=True)
