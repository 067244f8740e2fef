#!/usr/bin/env python3
"""one-off fuzz of hexvec_mlp_actions on the GPU box: random board sizes, batch sizes, pools, slot tables, skipped boards,
both kernels (matrix cores, one / two wavefronts per tile; general widths): logits against the torch fp32 forward,
actions against hexvec_sample_masked on the kernel's logits"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hex_gym_env_amd.vec_selfplay import MlpActorPolicy, mlp_actions, sample_masked

rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
t0, worst = time.time(), 0.0
for it in range(int(os.environ.get("ITERS", 300))):
    n = int(rng.integers(2, 20))
    C = n * n
    hidden = (64, 64) if rng.random() < 0.7 else tuple(int(16 * rng.integers(1, 9)) for _ in range(2))
    rows = int(rng.choice([1, 31, 32, 33, 63, 64, 65, 127, 500, 4097, 20000, 50000, 70000])) if rng.random() < 0.6 else int(rng.integers(1, 3000))
    if C > 200:
        rows = min(rows, 20000)
    P = int(rng.integers(1, 7))
    pols = [MlpActorPolicy(C, hidden).cuda() for _ in range(P)]
    with torch.no_grad():
        for p_ in pols:
            for w in p_.parameters():
                w.mul_(float(rng.uniform(0.5, 4.0)))
    table = torch.stack([p_.packed() for p_ in pols])
    g = torch.Generator(device="cuda").manual_seed(it)
    obs = (torch.randint(0, 3, (rows, C), generator=g, device="cuda") - 1).to(torch.int8)
    legal = (obs == 0).to(torch.uint8)
    n_slots = int(rng.integers(P, P + 5))
    of_slot = torch.tensor(rng.integers(0, P, n_slots), dtype=torch.int32, device="cuda")
    slot = torch.randint(-2, n_slots + 1, (rows,), generator=g, device="cuda")          # incl. out-of-range (clamped)
    active = (torch.rand(rows, generator=g, device="cuda") < rng.uniform(0.05, 1.0)).to(torch.uint8) if rng.random() < 0.7 else None
    use_slot = P > 1 or rng.random() < 0.5
    det = bool(rng.random() < 0.3)
    logits = torch.full((rows, C), float("nan"), device="cuda")
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    bad = torch.zeros((), dtype=torch.long, device="cuda")
    a = mlp_actions(table, C, hidden[0], hidden[1], obs, legal, row_active=active, slot=slot if use_slot else None, slot_bias=1,
                    policy_of_slot=of_slot if use_slot else None, deterministic=det, logits_out=logits, bad_rows=bad,
                    rng=(st, 99, it & 7, False))
    which = of_slot[(slot + 1).clamp(0, n_slots - 1).long()].long() if use_slot else torch.zeros(rows, dtype=torch.long, device="cuda")
    with torch.no_grad():
        ref = torch.stack([p_.action_net(p_.policy_net(obs.float())) for p_ in pols])[which, torch.arange(rows, device="cuda")]
    on = active.bool() if active is not None else torch.ones(rows, dtype=torch.bool, device="cuda")
    scale = float(ref[on].abs().max()) if bool(on.any()) else 1.0
    err = float((logits[on] - ref[on]).abs().max()) if bool(on.any()) else 0.0
    worst = max(worst, err / max(scale, 1.0))
    assert err <= 2e-5 * max(scale, 1.0), (it, n, hidden, rows, P, err, scale)
    assert bool(torch.isnan(logits[~on]).all()) and bool((a[~on] == 0).all())
    if det:
        want = logits.masked_fill(legal == 0, float("-inf")).argmax(1)
        ok = on & (legal.sum(1) > 0)
        assert torch.equal(a[ok].long(), want[ok]), (it, "argmax")
    else:
        b = sample_masked(logits.nan_to_num(0.0), legal, None, None, row_active=active, rng=(st, 99, it & 7, False))
        assert torch.equal(a, b), (it, "sample", n, hidden, rows, P)
    assert int(bad) == int(((legal.sum(1) == 0) & on).sum())
    if it % 25 == 0:
        print("iter %d ok (%dx%d, hidden %s, %d rows, %d networks; worst relative logit error so far %.2e; %.0f s)"
              % (it, n, n, hidden, rows, P, worst, time.time() - t0), flush=True)
print("fuzz ok: worst logit error %.2e of the logits' scale" % worst)
