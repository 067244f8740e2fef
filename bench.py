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
back inside ONE barrier + synchronize bracket; every repetition is timed with
HIP events on the launch stream.  ms_per_step is the MEDIAN repetition (max
over ranks) / K and `value` follows from it; wall_ms_per_step (bracket wall
time / all steps) is printed beside it as the cross-check.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_GBPS = 6290.0
PROFILE_ROUND = "r02"       # profiles/<round>/bench_pmc_summary.json: the PMC passes of the driver's command


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
                    help="skip the policy-in-the-loop leg (BASELINE config 3: 7x7, 32 768 envs, SelfplayWrapper + MLP policy)")
    ap.add_argument("--layout", choices=["auto", "rollout", "bitboard"], default="auto",
                    help="state layout of the rollout handles (include/hexvec.h HEXVEC_F_ROLLOUT_LAYOUT); "
                         "auto = by the number of wavefronts the geometry gives (HexVecEnv rollout_layout='auto')")
    ap.add_argument("--reps", type=int, default=0, help="timed repetitions of the --steps region (0 = auto, >= 20)")
    ap.add_argument("--warmup-internal-ms", type=float, default=150.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import hex_gym_env_amd as hx
    from hex_gym_env_amd import dist as hxdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if under_launcher:      # torch.distributed.run: one process per GPU, RCCL ("nccl" on ROCm)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    N, E, T = args.board, args.envs, max(1, min(args.chunk, args.steps))
    C = N * N
    # the rollout handle keeps its state in the rollout layout (HEXVEC_F_ROLLOUT_LAYOUT: lane-per-board
    # union-find); --layout bitboard runs the same rollout on the default (host-stepping) layout
    env = hx.HexVecEnv(N, E, device=dev, encoding="B", seed=0, env_offset=rank * E,
                       rollout_layout={"auto": "auto", "rollout": True, "bitboard": False}[args.layout])
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
    reps = args.reps or int(min(2000, max(20, 60.0 / max(e0.elapsed_time(e1), 1e-3))))
    reps = int(hxdist.max_over_ranks(reps, dev))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    fence()
    t0 = time.perf_counter()
    evs[0].record()         # libhexvec launches on torch's current stream, as do these events
    for r in range(reps):
        run_steps(args.steps)
        evs[r + 1].record()
    fence()
    wall = time.perf_counter() - t0
    rep_ms = sorted(evs[r].elapsed_time(evs[r + 1]) for r in range(reps))
    med_ms = rep_ms[len(rep_ms) // 2] if reps % 2 else 0.5 * (rep_ms[reps // 2 - 1] + rep_ms[reps // 2])
    med_ms = hxdist.max_over_ranks(med_ms, dev)
    wall = hxdist.max_over_ranks(wall, dev)
    ms_per_step = med_ms / args.steps
    value = world * E / (ms_per_step * 1e-3)

    # the dominant kernel: bytes it must write per env step (obs + mask + action + reward + done)
    bytes_per_step = 2 * C + 9
    launches = launches_per_rep
    # roofline: the AVERAGE launch duration over the timed region (HIP events), as the contract asks;
    # ms_per_step / value use the median repetition
    mean_ms = hxdist.max_over_ranks(sum(rep_ms) / reps, dev)
    avg_launch_s = mean_ms * 1e-3 / launches
    steps_per_launch = args.steps / launches
    achieved = E * steps_per_launch * bytes_per_step / avg_launch_s / 1e9

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
                       "source": f"profiles/{PROFILE_ROUND}/{os.path.basename(pfile)} (rocprofv3 --pmc, same command)"}
    except Exception:
        traffic, on_chip = None, None

    extra = {}
    if not args.no_host_stepped:
        # one hexvec_step launch per env step on a default handle (bitboard state: 144 B per board at 11x11)
        envs = hx.HexVecEnv(N, E, device=dev, encoding="B", seed=0, env_offset=rank * E)
        ks = 1000
        for _ in range(300):
            envs.step()
        fence()
        h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        h0.record()
        for _ in range(ks):
            envs.step()
        h1.record()
        fence()
        hs_wall = time.perf_counter() - t0
        extra["host_stepped"] = {"env_steps_per_s": world * E * ks / hs_wall,
                                 "us_per_launch_stream": 1e3 * h0.elapsed_time(h1) / ks,
                                 "bytes_per_step": 3 * C + 9, "kernel": envs.last_kernel(),
                                 "achieved_GBps": E * ks * (3 * C + 9) / hs_wall / 1e9}
        del envs

    # W2: two-sided device-resident rollout, every ply recorded (SURVEY 8(d) unit W2)
    if not args.no_w2:
        del out
        env2 = hx.HexVecEnv(N, E, device=dev, encoding="B", two_sided=True, seed=1, env_offset=rank * E,
                            rollout_layout={"auto": "auto", "rollout": True, "bitboard": False}[args.layout])
        T2 = 64
        out2 = dict(obs=torch.empty((T2, E, C), dtype=torch.int8, device=dev),
                    legal=torch.empty((T2, E, C), dtype=torch.uint8, device=dev),
                    actions=torch.empty((T2, E), dtype=torch.int32, device=dev),
                    reward=torch.empty((T2, E), dtype=torch.float32, device=dev),
                    done=torch.empty((T2, E), dtype=torch.uint8, device=dev))
        for _ in range(60):                      # ~12 ms: past the clock ramp after start-up
            env2.rollout_random(T2, out2)
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nl = 40
        for _ in range(nl):
            env2.rollout_random(T2, out2)
        e1.record()
        fence()
        ms = e0.elapsed_time(e1) / nl
        plies = E * T2 / (ms * 1e-3)
        extra["rollout_w2"] = {"plies_per_s": world * plies, "avg_launch_ms": ms,
                               "bytes_per_ply": 2 * C + 9,
                               "achieved_GBps": plies * (2 * C + 9) / 1e9,
                               "frac": plies * (2 * C + 9) / 1e9 / HBM_PEAK_GBPS,
                               "kernel": env2.last_kernel()}

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
            for _ in range(200):
                run()
            fence()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            kc = 500
            g0.record()
            for _ in range(kc):
                run()
            g1.record()
            fence()
            us = 1e3 * g0.elapsed_time(g1) / kc
            extra["collect_step"] = {"workload": "7x7, 32768 envs, SelfplayWrapper semantics, MLP 64-64 agent and opponents "
                                                 "(20-slot pool), one hipGraph per collection step",
                                     "us_per_step": us, "agent_steps_per_s": E3 / (us * 1e-6),
                                     "degenerate_rows": env3.degenerate_rows(agent)}
            del env3, run
        except Exception as e:  # the secondary leg never takes the headline down with it
            extra["collect_step"] = {"error": repr(e)}

    # episode statistics: the only cross-GPU exchange (logging), one all-gather over RCCL
    per_rank, stats = hxdist.gather_episode_stats(env.episode_stats())
    stats = stats.cpu().tolist()

    if rank == 0:
        line = {
            "metric": "env-steps/sec (random policy, 11x11)" if N == 11
                      else f"env-steps/sec (random policy, {N}x{N})",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "timed_reps": reps,
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
        line.update(extra)
        if on_chip:
            line["on_chip"] = on_chip
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(N, seconds=args.cpu_seconds)
            line["cpu_baseline_1thread"] = cpu_baseline(N, seconds=min(3.0, args.cpu_seconds), cores=1)
        print(json.dumps(line), flush=True)
    if under_launcher:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
