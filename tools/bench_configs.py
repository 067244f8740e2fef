#!/usr/bin/env python3
"""bench.py lines for the BASELINE geometries that are not the headline (run on the GPU box):
7x7 / 32 768 envs (config 3's env), 13x13 / 16 384 envs (config 5's per-GPU share), 13x13 / 65 536
and the headline 11x11 / 65 536 -- W1 rollout on both state layouts, host-stepped step, W2 rollout.
Prints one compact row per geometry and layout; the full JSON lines go to gpurun_out/cfg_*.json."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
steps = sys.argv[1] if len(sys.argv) > 1 else "200"
for n, e in ((7, 32768), (13, 16384), (13, 65536), (11, 65536)):
    for lay in ("rollout", "bitboard"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--board", str(n), "--envs", str(e), "--steps", steps,
               "--warmup", "5", "--no-cpu", "--no-collect", "--layout", lay]
        if lay == "bitboard":
            cmd += ["--no-host-stepped"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, e, lay, "FAILED", out.stderr[-500:])
            continue
        d = json.loads(line[0])
        open(os.path.join(ROOT, "gpurun_out", f"cfg_{n}_{e}_{lay}_{steps}.json"), "w").write(line[0] + "\n")
        r = d["roofline"]
        row = "%2dx%-2d E=%6d %-8s T=%s: W1 %6.2f G steps/s  %5.0f GB/s frac %.3f (obs-only %.3f) [%s]" % (
            n, n, e, lay, steps, d["value"] / 1e9, r["achieved"], r["frac"], r["obs_only_frac"], r["kernel"])
        if "rollout_w2" in d:
            w = d["rollout_w2"]
            row += " | W2 %6.2f G plies/s frac %.3f" % (w["plies_per_s"] / 1e9, w["frac"])
        if "host_stepped" in d:
            h = d["host_stepped"]
            row += " | host-stepped %5.2f G steps/s, %.1f us/launch, %4.0f GB/s" % (
                h["env_steps_per_s"] / 1e9, h["us_per_launch_stream"], h["achieved_GBps"])
        print(row, flush=True)
