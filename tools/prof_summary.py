#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/profile_bench.sh:

    python3 tools/prof_summary.py gpurun_out/prof_<tag> [--config board,envs,steps_per_launch]

-> <dir>/kernel_stats.csv  (copy of rocprofv3's --stats table of the trace pass)
   <dir>/pmc_summary.json  (per-kernel mean of every counter over the dispatches of the PMC passes)
and a short text report on stdout.  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them
(FETCH_SIZE is half-counted on gfx950: double it when comparing with bytes)."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def mode_of(name):
    m = re.search(r"k_(qpb|lpb|hex)<\s*(\d+),\s*(\d+),\s*(\d+)", name)
    if not m:
        return None
    fam, n, enc, mode = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
    names = {0: "apply", 1: "rollout2", 2: "step", 3: "reset", 4: "rollout1", 5: "set_state", 6: "set_boards", 7: "step", 8: "step_agent", 9: "step_opponent"}
    return fam, n, enc, names.get(mode, str(mode))


def main():
    d = sys.argv[1]
    cfg = None
    if "--config" in sys.argv:
        b, e, t = (int(x) for x in sys.argv[sys.argv.index("--config") + 1].split(","))
        cfg = {"board": b, "envs": e, "steps_per_launch": t}
    if cfg is None and os.path.exists(os.path.join(d, "trace.json")):   # the bench line of the trace pass names it
        try:
            line = [l for l in open(os.path.join(d, "trace.json")) if l.startswith("{")][-1]
            c = json.loads(line)["config"]
            cfg = {"board": c["board_size"], "envs": c["envs_per_gpu"], "steps_per_launch": c["steps_per_launch"]}
        except Exception:
            cfg = None
    stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    report = []
    if stats:
        shutil.copy(stats[0], os.path.join(d, "kernel_stats.csv"))
        for r in csv.DictReader(open(stats[0])):
            report.append("%-70s calls %6s avg %10.1f us  min %10.1f  max %10.1f" % (
                r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    sums = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    kname = {}
    rows = []
    for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f))]
    # the board size of the bench's main workload = that of its rollout kernel; kernels of other sizes (the
    # policy-in-the-loop leg runs 7x7) are kept apart as "<mode>@NxN"
    sizes = [mode_of(r["Kernel_Name"]) for r in rows]
    main_n = next((m[1] for m in sizes if m and m[3] == "rollout1"), None)
    if main_n is None:
        main_n = next((m[1] for m in sizes if m), None)
    for r in rows:
        if True:
            m = mode_of(r["Kernel_Name"])
            if not m:
                continue
            key = m[3] if m[1] == main_n else "%s@%dx%d" % (m[3], m[1], m[1])
            kname[key] = "k_%s<%d,%d,%s>" % (m[0], m[1], m[2], m[3])
            acc = sums[key][r["Counter_Name"]]
            acc[0] += float(r["Counter_Value"])
            acc[1] += 1
    out = {"config": cfg, "kernel": kname.get("rollout1", ""), "kernels": kname, "counters": {}}
    for key, cs in sums.items():
        out["counters"][key] = {c: v[0] / v[1] for c, v in sorted(cs.items())}
        out["counters"][key]["dispatches_per_pass"] = max(v[1] for v in cs.values())
    json.dump(out, open(os.path.join(d, "pmc_summary.json"), "w"), indent=1)
    print("\n".join(report))
    for key, c in out["counters"].items():
        w = c.get("SQ_WAVES", 0)
        line = "%s (%s): waves %d" % (key, kname[key], w)
        if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
            line += "  HBM traffic %.1f MB (write %.1f + 2 x fetch %.1f)" % (
                (c["WRITE_SIZE"] + 2 * c["FETCH_SIZE"]) * 1024 / 1e6, c["WRITE_SIZE"] * 1024 / 1e6, c["FETCH_SIZE"] * 1024 / 1e6)
        if w:
            line += "  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_WR %.0f" % (
                c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w,
                c.get("SQ_INSTS_VMEM_WR", 0) / w)
        if c.get("SQ_WAVE_CYCLES"):
            line += "  WAIT_ANY/WAVE_CYCLES %.2f" % (c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"])
            line += "  ACTIVE_VALU/WAVE_CYCLES %.2f" % (c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"])
        print(line)


if __name__ == "__main__":
    main()
