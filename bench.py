#!/usr/bin/env python3
"""bench.py -- env-steps/s of the vectorised Hex hot path on MI355X.

Workload (BASELINE.json configs[1]): 11x11, 65 536 parallel envs per GPU,
random policy.  One "step" = one SelfPlayEnv.step per env (agent ply +
opponent reply + auto reset; SURVEY 8(d) unit W1), agent and opponent both
uniform-legal, drawn on the device.  The timed region runs the K steps as
launches of the device-resident self-play rollout kernel (k_lpb<rollout1>,
T = min(--chunk, K) steps per launch, every step's obs / mask / action /
reward / done written to HBM); the host-stepped rate (one hexvec_step launch
per step) is reported beside it.

Timing: a fixed internal warm-up of rollout launches (>= 150 ms, the chip needs
tens of ms of load to settle its clock; reported as warmup_internal_ms) precedes the
--warmup steps.  The K-step region is then repeated `timed_reps` times back to
back inside ONE barrier + synchronize bracket, timed with HIP events on the launch
stream: one event per group of `reps_per_event` repetitions (~1 ms of launches; an
event between every two launches would add its own ~3 us barrier packet to each).
ms_per_step is the MEDIAN group (max over ranks) / repetitions / K and `value` follows from it; wall_ms_per_step (bracket wall
time / all steps) is printed beside it as the cross-check.  Every secondary leg
(host_stepped, rollout_w2, collect_step, ppo_update) is measured the same way: >= 150 ms of
warm-up on that leg's own kernels, >= 20 repetitions, median / min / max reported.

After the timed region (outside it) one more launch of the same kernel at the same geometry runs on a fresh
seeded handle and its first rows are compared with the CPU oracle: `verified` on the line; a mismatch exits 1.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher environment starts the N ranks itself (torch.distributed.run as a CHILD
process, decided before anything touches a GPU) and relays rank 0's line.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_GBPS = 6290.0
PROFILE_ROUND = "r03"       # profiles/<round>/bench_pmc_summary.json: the PMC passes of the driver's command
LEG_WARM_MS = 150.0         # warm-up of every secondary leg on its own kernels
LEG_REPS = 20


# --------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the ranks as a child process
# --------------------------------------------------------------------------------------------------
def launcher_command(nranks, argv, port):
    """what the driver runs for N > 1 (one rank per GPU over RCCL), with this bench's own arguments"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv, run=subprocess.run, visible=None):
    """Start args.gpus ranks under torch.distributed.run as a CHILD process (never an exec; nothing in this process
    has touched a GPU yet: torch.cuda.device_count() does not initialise one), relay rank 0's single JSON line and
    return the exit code.  Refuses when fewer GPUs are visible than asked for."""
    n = args.gpus
    if visible is None:
        if args.dry_ranks:
            visible = n
        else:
            import torch
            visible = torch.cuda.device_count()
    if visible < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, found {visible}; not reporting a smaller job under "
              f"that name", file=sys.stderr)
        return 2
    child_argv = [a for a in argv if a != "--force-launcher"]
    cmd = launcher_command(n, child_argv, _free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = [l for l in (p.stdout or "").splitlines() if l.startswith("{")]
    other = [l for l in (p.stdout or "").splitlines() if not l.startswith("{")]
    if other:
        print("\n".join(other), file=sys.stderr)
    if p.returncode != 0:
        print(f"bench.py: the {n}-rank launch failed with exit code {p.returncode}", file=sys.stderr)
        for l in lines:
            print(l, flush=True)        # e.g. a line whose `verified` block says false
        return p.returncode
    if len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 3
    got = json.loads(lines[0]).get("n_gpus")
    if got != n:
        print(f"bench.py: asked for {n} GPUs, the ranks report n_gpus = {got}", file=sys.stderr)
        return 3
    print(lines[0], flush=True)
    return 0


def dry_ranks(args):
    """--dry-ranks (CPU test of the launch plumbing): the rank logic of main() -- process group, barrier,
    max-over-ranks, the stats all_gather, rank 0's line -- over gloo, with no GPU and no env behind it"""
    import torch
    import torch.distributed as dist
    from hex_gym_env_amd import dist as hxdist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    hxdist.barrier()
    ms = hxdist.max_over_ranks(1.0 + rank)
    per, tot = hxdist.gather_episode_stats(torch.tensor([1.0, 2.0, 3.0, float(rank)], dtype=torch.float64))
    if rank == 0:
        print(json.dumps({"metric": "dry run of the rank plumbing (no GPU work)", "value": 0.0, "n_gpus": world,
                          "dry": True, "max_over_ranks": ms, "stats_total": tot.tolist(),
                          "config": {"envs_per_gpu": args.envs, "global_envs": world * args.envs}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


# --------------------------------------------------------------------------------------------------
def cpu_baseline(board, seconds=12.0, cores=None):
    """The oracle (a C port of the reference algorithm) on this host's cores, same workload,
    run in chunks until `seconds` of CPU work have been timed."""
    from oracle import hex_oracle as ho
    cores = cores or os.cpu_count() or 1
    E = 4096 if cores > 1 else 256
    o = ho.OracleVec(board, E, seed=0, flags=ho.F_ENC_B | ho.F_AGENT_RANDOM)
    o.set_threads(cores)
    o.step_repeat(50)                      # thread start-up, page faults
    K, chunk = 0, 2000 if cores > 1 else 500
    t0 = time.perf_counter()
    while True:
        o.step_repeat(chunk)
        K += chunk
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"value": E * K / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{E} envs x {K} self-play steps, {board}x{board}, oracle/hex_oracle.c "
                      f"on {cores} threads, {dt:.1f} s"}


def verify_against_oracle(hx, torch, dev, N, E, T, layout, rank, out, envs=2048):
    """One launch of the timed kernel at the timed geometry on a FRESH handle (same seed, same env_offset as the timed
    one at its creation), outside the timed region; the first `envs` boards' recorded rows (obs, legal mask, action,
    reward, done of every step) against the CPU oracle started at the same global env ids.  The oracle is the checker
    here, as in tests/ (SelfplayWrapper.py:69-89, :174-199 through oracle/hex_oracle.c)."""
    import numpy as np
    from oracle import hex_oracle as ho
    nv = min(envs, E)
    env = hx.HexVecEnv(N, E, device=dev, encoding="B", seed=0, env_offset=rank * E, rollout_layout=layout)
    env.rollout_selfplay(T, out)
    kernel = env.last_kernel()
    torch.cuda.synchronize(dev)
    o = ho.OracleVec(N, nv, seed=0, flags=ho.F_ENC_B | ho.F_AGENT_RANDOM, env_offset=rank * E)
    o.set_threads(min(os.cpu_count() or 1, 64))
    ref = dict(zip(("obs", "legal", "actions", "reward", "done"), o.rollout_selfplay(T)))
    bad = [k for k in ref if not np.array_equal(out[k][:T, :nv].cpu().numpy(), ref[k])]
    del env
    return {"envs": nv, "steps": T, "equal": not bad, "kernel": kernel, "rows": sorted(ref),
            **({"mismatch": bad} if bad else {})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--board", type=int, default=11)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--chunk", type=int, default=200, help="env steps per rollout launch")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="length of the CPU-baseline sample")
    ap.add_argument("--no-host-stepped", action="store_true")
    ap.add_argument("--no-w2", action="store_true", help="skip the two-sided rollout leg")
    ap.add_argument("--no-collect", action="store_true",
                    help="skip the policy-in-the-loop legs (BASELINE config 3: 7x7, 32 768 envs, SelfplayWrapper + MLP "
                         "policy: collection step and PPO update)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-timing oracle check of the timed kernel")
    ap.add_argument("--layout", choices=["auto", "rollout", "bitboard"], default="auto",
                    help="state layout of the rollout handles (include/hexvec.h HEXVEC_F_ROLLOUT_LAYOUT); "
                         "auto = by the number of wavefronts the geometry gives (HexVecEnv rollout_layout='auto')")
    ap.add_argument("--reps", type=int, default=0, help="timed event groups of the --steps region (0 = auto, >= 20)")
    ap.add_argument("--event-group", type=int, default=0,
                    help="repetitions between two HIP events of the timed region (0 = auto: ~1 ms of launches)")
    ap.add_argument("--warmup-internal-ms", type=float, default=150.0)
    ap.add_argument("--force-launcher", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (tests the self-launch path)")
    ap.add_argument("--dry-ranks", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and (args.gpus > 1 or args.force_launcher):
        sys.exit(self_launch(args, sys.argv[1:]))       # before torch.cuda / any handle exists in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:      # never report a job of another size under this --gpus
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE {world} ranks", file=sys.stderr)
        sys.exit(2)
    if args.dry_ranks:
        sys.exit(dry_ranks(args))

    import torch
    import torch.distributed as dist
    import hex_gym_env_amd as hx
    from hex_gym_env_amd import dist as hxdist

    if under_launcher:      # torch.distributed.run: one process per GPU, RCCL ("nccl" on ROCm)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.device_count() <= local:
            print(f"bench.py: rank {rank} has no GPU {local}", file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    N, E, T = args.board, args.envs, max(1, min(args.chunk, args.steps))
    C = N * N
    layout = {"auto": "auto", "rollout": True, "bitboard": False}[args.layout]
    # the rollout handle keeps its state in the rollout layout (HEXVEC_F_ROLLOUT_LAYOUT: lane-per-board
    # union-find); --layout bitboard runs the same rollout on the default (host-stepping) layout
    env = hx.HexVecEnv(N, E, device=dev, encoding="B", seed=0, env_offset=rank * E, rollout_layout=layout)
    out = dict(obs=torch.empty((T, E, C), dtype=torch.int8, device=dev),
               legal=torch.empty((T, E, C), dtype=torch.uint8, device=dev),
               actions=torch.empty((T, E), dtype=torch.int32, device=dev),
               reward=torch.empty((T, E), dtype=torch.float32, device=dev),
               done=torch.empty((T, E), dtype=torch.uint8, device=dev))

    def run_steps(k):
        """k env steps as rollout launches of <= T steps; returns the launch count"""
        n = 0
        while k > 0:
            t = min(T, k)
            env.rollout_selfplay(t, out)
            k -= t
            n += 1
        return n

    def fence():
        torch.cuda.synchronize(dev)
        hxdist.barrier()
        torch.cuda.synchronize(dev)

    def timed_leg(fn, inner, reps=LEG_REPS, warm_ms=LEG_WARM_MS):
        """fn() = one launch (or one graph replay) of a secondary leg.  >= warm_ms of warm-up on fn itself, then
        `reps` repetitions of `inner` calls, each repetition bracketed by HIP events on the launch stream.
        -> per-call microseconds: median, min, max over the repetitions, and the wall time of the whole region."""
        t0 = time.perf_counter()
        while True:
            for _ in range(inner):
                fn()
            torch.cuda.synchronize(dev)
            if 1e3 * (time.perf_counter() - t0) >= warm_ms:
                break
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        fence()
        w0 = time.perf_counter()
        ev[0].record()
        for r in range(reps):
            for _ in range(inner):
                fn()
            ev[r + 1].record()
        fence()
        wall = time.perf_counter() - w0
        us = sorted(1e3 * ev[r].elapsed_time(ev[r + 1]) / inner for r in range(reps))
        med = us[reps // 2] if reps % 2 else 0.5 * (us[reps // 2 - 1] + us[reps // 2])
        return {"us_median": med, "us_min": us[0], "us_max": us[-1], "reps": reps, "calls_per_rep": inner,
                "wall_us_per_call": 1e6 * wall / (reps * inner)}

    # internal clock-ramp warm-up, independent of --warmup
    t0 = time.perf_counter()
    while True:
        for _ in range(8):
            env.rollout_selfplay(T, out)
        torch.cuda.synchronize(dev)
        warm_ms = 1e3 * (time.perf_counter() - t0)
        if warm_ms >= args.warmup_internal_ms:
            break
    run_steps(args.warmup)
    main_kernel = env.last_kernel()
    torch.cuda.synchronize(dev)
    # size the repetition count from one probe region: >= 20 repetitions and >= ~60 ms of timed work
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launches_per_rep = run_steps(args.steps)
    e1.record()
    torch.cuda.synchronize(dev)
    probe_ms = max(e0.elapsed_time(e1), 1e-3)
    # An event between two launches costs the stream ~3 us (a barrier packet of its own: 61.6 us launch to launch against
    # 58.2 us without, at 20 steps per launch) -- the instrument, not the workload.  One event per GROUP of repetitions
    # that lasts >= ~1 ms keeps it below 0.5 %; >= 20 groups and >= ~60 ms of timed work.
    group = args.event_group or int(min(64, max(1, round(1.0 / probe_ms))))
    group = int(hxdist.max_over_ranks(group, dev))
    groups = args.reps or int(min(2000, max(20, 60.0 / (group * probe_ms))))
    groups = int(hxdist.max_over_ranks(groups, dev))
    reps = groups * group
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
    fence()
    t0 = time.perf_counter()
    evs[0].record()         # libhexvec launches on torch's current stream, as do these events
    for g in range(groups):
        for _ in range(group):
            run_steps(args.steps)
        evs[g + 1].record()
    fence()
    wall = time.perf_counter() - t0
    rep_ms = sorted(evs[g].elapsed_time(evs[g + 1]) / group for g in range(groups))
    med_ms = rep_ms[groups // 2] if groups % 2 else 0.5 * (rep_ms[groups // 2 - 1] + rep_ms[groups // 2])
    med_ms = hxdist.max_over_ranks(med_ms, dev)
    wall = hxdist.max_over_ranks(wall, dev)
    ms_per_step = med_ms / args.steps
    value = world * E / (ms_per_step * 1e-3)

    # the dominant kernel: bytes it must write per env step (obs + mask + action + reward + done)
    bytes_per_step = 2 * C + 9
    launches = launches_per_rep
    # roofline: the AVERAGE launch duration over the timed region (HIP events), as the contract asks;
    # ms_per_step / value use the median repetition
    mean_ms = hxdist.max_over_ranks(sum(rep_ms) / groups, dev)
    avg_launch_s = mean_ms * 1e-3 / launches
    steps_per_launch = args.steps / launches
    achieved = E * steps_per_launch * bytes_per_step / avg_launch_s / 1e9

    # the timed kernel at the timed geometry against the oracle, outside the timed region (every rank checks its shard)
    verified = None
    if not args.no_verify:
        verified = verify_against_oracle(hx, torch, dev, N, E, T, layout, rank, out)
        ok = hxdist.max_over_ranks(0.0 if verified["equal"] else 1.0, dev) == 0.0
        verified["all_ranks_equal"] = ok

    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    # (FETCH_SIZE is half-counted on gfx950, MI355X_MICROARCH.md HBM section); null if the
    # profile was taken for another configuration or kernel
    traffic, on_chip = None, None
    try:
        pdir = os.path.join(ROOT, "profiles", PROFILE_ROUND)
        pfile = os.path.join(pdir, f"bench_T{T}_pmc_summary.json")        # e.g. the 200-steps-per-launch passes
        if not os.path.exists(pfile):
            pfile = os.path.join(pdir, "bench_pmc_summary.json")            # the driver's command
        prof = json.load(open(pfile))
        if (prof["config"] == {"board": N, "envs": E, "steps_per_launch": T}
                and prof["kernel"].startswith(main_kernel.split("<")[0])):
            c = prof["counters"]["rollout1"]
            traffic = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            glob_insts = c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]
            lds = c.get("SQ_INSTS_LDS", 0.0)
            on_chip = {"lds_insts_per_launch": lds, "global_insts_per_launch": glob_insts,
                       "lds_share_of_memory_insts": lds / max(lds + glob_insts, 1.0),
                       "lds_bank_conflict_cycle_frac": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
                                                        if c.get("SQ_LDS_IDX_ACTIVE") else None),
                       "valu_insts_per_launch": c["SQ_INSTS_VALU"], "salu_insts_per_launch": c["SQ_INSTS_SALU"],
                       "waves_per_launch": c["SQ_WAVES"],
                       "source": f"profiles/{PROFILE_ROUND}/{os.path.basename(pfile)} (rocprofv3 --pmc, same command, "
                                 f"the builder's box; not measured in this run)"}
    except Exception:
        traffic, on_chip = None, None

    extra = {}
    if not args.no_host_stepped:
        # one hexvec_step launch per env step on a default handle (bitboard state: 144 B per board at 11x11)
        envs = hx.HexVecEnv(N, E, device=dev, encoding="B", seed=0, env_offset=rank * E)
        leg = timed_leg(envs.step, inner=100)
        us = hxdist.max_over_ranks(leg["us_median"], dev)
        extra["host_stepped"] = {"env_steps_per_s": world * E / (leg["wall_us_per_call"] * 1e-6),
                                 "us_per_launch_stream": us, "us_min_max": [leg["us_min"], leg["us_max"]],
                                 "reps": leg["reps"], "launches_per_rep": leg["calls_per_rep"],
                                 "bytes_per_step": 3 * C + 9, "kernel": envs.last_kernel(),
                                 "achieved_GBps": E * (3 * C + 9) / (us * 1e-6) / 1e9,
                                 "frac": E * (3 * C + 9) / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS}
        del envs

    # W2: two-sided device-resident rollout, every ply recorded (SURVEY 8(d) unit W2)
    if not args.no_w2:
        del out
        env2 = hx.HexVecEnv(N, E, device=dev, encoding="B", two_sided=True, seed=1, env_offset=rank * E,
                            rollout_layout=layout)
        T2 = 64
        out2 = dict(obs=torch.empty((T2, E, C), dtype=torch.int8, device=dev),
                    legal=torch.empty((T2, E, C), dtype=torch.uint8, device=dev),
                    actions=torch.empty((T2, E), dtype=torch.int32, device=dev),
                    reward=torch.empty((T2, E), dtype=torch.float32, device=dev),
                    done=torch.empty((T2, E), dtype=torch.uint8, device=dev))
        leg = timed_leg(lambda: env2.rollout_random(T2, out2), inner=4)
        us = hxdist.max_over_ranks(leg["us_median"], dev)
        plies = E * T2 / (us * 1e-6)
        extra["rollout_w2"] = {"plies_per_s": world * plies, "avg_launch_ms": us * 1e-3,
                               "launch_us_min_max": [leg["us_min"], leg["us_max"]], "reps": leg["reps"],
                               "launches_per_rep": leg["calls_per_rep"], "plies_per_launch": T2,
                               "bytes_per_ply": 2 * C + 9,
                               "achieved_GBps": plies * (2 * C + 9) / 1e9,
                               "frac": plies * (2 * C + 9) / 1e9 / HBM_PEAK_GBPS,
                               "kernel": env2.last_kernel()}
        del env2, out2

    # BASELINE config 3, policy in the loop (secondary, N = 1 only): one PPO-style collection step = the agent's
    # actor network + masked sampling + SelfPlayEnv.step with MLP opponents from a 20-slot pool, every board its
    # own opponent slot, captured as one hipGraph.  fp32 networks with random-init weights of SB3's default shape.
    if not args.no_collect and world == 1:
        try:
            from hex_gym_env_amd.vec_selfplay import VecSelfPlayEnv, MlpActorPolicy
            n3, E3 = 7, 32768
            torch.manual_seed(0)
            agent, opp = MlpActorPolicy(n3 * n3).to(dev), MlpActorPolicy(n3 * n3).to(dev)
            env3 = VecSelfPlayEnv(board_size=n3, num_envs=E3, buffer_size=20, base_model=opp, seed=1, device=dev)
            env3.reset()
            run = env3.capture_collect_step(agent)
            leg = timed_leg(run, inner=50)
            us = leg["us_median"]
            extra["collect_step"] = {"workload": "7x7, 32768 envs, SelfplayWrapper semantics, MLP 64-64 agent and opponents "
                                                 "(20-slot pool), one hipGraph per collection step",
                                     "us_per_step": us, "us_min_max": [leg["us_min"], leg["us_max"]],
                                     "reps": leg["reps"], "replays_per_rep": leg["calls_per_rep"],
                                     "agent_steps_per_s": E3 / (us * 1e-6),
                                     "degenerate_rows": env3.degenerate_rows(agent)}
            del run
            # the same collection as a PPO collector runs it between two updates: several steps per hipGraph, every step's
            # rows written straight into the rollout buffer (VecSelfPlayEnv.capture_collect): no inter-graph gap and no
            # host call per step
            NS = 8
            runn = env3.capture_collect(agent, NS)
            leg = timed_leg(runn, inner=8)
            extra["collect_rollout"] = {"workload": f"the collect_step workload, {NS} steps per hipGraph, rows written "
                                                    f"straight into a [{NS}, E, ...] rollout buffer",
                                        "steps_per_graph": NS, "us_per_step": leg["us_median"] / NS,
                                        "us_min_max": [leg["us_min"] / NS, leg["us_max"] / NS], "reps": leg["reps"],
                                        "agent_steps_per_s": E3 * NS / (leg["us_median"] * 1e-6),
                                        "degenerate_rows": env3.degenerate_rows(agent)}
            del runn
            # the PPO update on what such steps collect (reported apart from the env metric):
            # plain torch fp32, SB3's default MlpPolicy shape (separate 64-64 tanh actor and critic), clipped
            # surrogate + value + entropy loss with the legal mask, Adam, one minibatch of one collected step
            # (32 768 samples) per optimizer step.  The buffer rows are synthetic (what the collection step above
            # returns, held fixed): the arithmetic does not depend on their values.
            from hex_gym_env_amd.vec_selfplay import ppo_minibatch_step, MlpCritic
            critic = MlpCritic(n3 * n3).to(dev)
            opt = torch.optim.Adam(list(agent.parameters()) + list(critic.parameters()), lr=3e-4)
            obs3 = env3.vec.obs.reshape(E3, -1).float().clone()
            legal3 = env3.vec.legal.bool().clone()
            with torch.no_grad():
                lg = agent.masked_logits(obs3, legal3)
                act3 = torch.distributions.Categorical(logits=lg).sample()
                oldlp = torch.log_softmax(lg, -1).gather(1, act3[:, None]).squeeze(1)
            adv3, ret3 = torch.randn(E3, device=dev), torch.randn(E3, device=dev)
            leg = timed_leg(lambda: ppo_minibatch_step(agent, critic, opt, obs3, legal3, act3, oldlp, adv3, ret3),
                            inner=5)
            extra["ppo_update"] = {"workload": "7x7, one minibatch = 32768 samples (one collected step), MLP 64-64 actor "
                                               "+ critic, clipped PPO loss with legal mask, Adam, fp32, plain torch",
                                   "us_per_minibatch": leg["us_median"], "us_min_max": [leg["us_min"], leg["us_max"]],
                                   "reps": leg["reps"], "samples_per_s": E3 / (leg["us_median"] * 1e-6),
                                   "note": "reported apart from the env metric (SURVEY 8(d) item 3)"}
            del env3
        except Exception as e:  # the secondary legs never take the headline down with them
            extra.setdefault("collect_step", {"error": repr(e)})
            if "error" not in extra["collect_step"]:
                extra["ppo_update"] = {"error": repr(e)}

    # episode statistics: the only cross-GPU exchange (logging), one all-gather over RCCL
    per_rank, stats = hxdist.gather_episode_stats(env.episode_stats())
    stats = stats.cpu().tolist()

    rc = 0
    if rank == 0:
        line = {
            "metric": "env-steps/sec (random policy, 11x11)" if N == 11
                      else f"env-steps/sec (random policy, {N}x{N})",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "timed_reps": reps, "reps_per_event": group,
            "warmup_internal_ms": warm_ms, "wall_ms_per_step": 1e3 * wall / (reps * args.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{N}x{N} Hex, {E} envs/GPU, SelfPlayEnv.step with uniform-legal "
                                   f"agent and opponent (device RNG), auto reset, "
                                   f"{T} steps per rollout launch",
                       "board_size": N, "envs_per_gpu": E, "global_envs": world * E,
                       "steps_per_launch": T, "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "obs_only_frac": achieved * C / bytes_per_step / HBM_PEAK_GBPS,
                         "kernel": main_kernel, "bytes_per_env_step": bytes_per_step,
                         "avg_launch_ms": avg_launch_s * 1e3, "launches": launches * reps,
                         "rep_ms_min_med_mean_max": [rep_ms[0], med_ms, mean_ms, rep_ms[-1]],
                         "frac_of_copy_ceiling": achieved / HBM_COPY_GBPS},
            "episodes": {"sum_return": stats[0], "sum_len": stats[1], "n": stats[2],
                         "agent_wins": stats[3]},
        }
        if verified is not None:
            line["verified"] = verified
        line.update(extra)
        if on_chip:
            line["on_chip"] = on_chip
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(N, seconds=args.cpu_seconds)
            line["cpu_baseline_1thread"] = cpu_baseline(N, seconds=min(3.0, args.cpu_seconds), cores=1)
        print(json.dumps(line), flush=True)
    if verified is not None and not verified["all_ranks_equal"]:
        if not verified["equal"]:
            print(f"bench.py: rank {rank}: the timed kernel's rows differ from the oracle: {verified}", file=sys.stderr)
        rc = 1
    if under_launcher:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
