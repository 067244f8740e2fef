#!/usr/bin/env python3
"""per-kernel means of a rocprofv3 --pmc pass (counter_collection.csv under the given directory) for the kernels whose
name contains one of the given substrings:  python3 tools/pmc_mfma_summary.py <dir> k< k_mlp"""
import collections, csv, glob, sys
d, pats = sys.argv[1], sys.argv[2:]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f[0])):
    if not any(p in r["Kernel_Name"] for p in pats):
        continue
    k = "%s grid %s" % (r["Kernel_Name"].split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", "?")))
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    c = {n: v[0] / v[1] for n, v in cs.items()}
    w = max(c.get("SQ_WAVES", 1), 1)
    print("%-44s waves %5d | per wavefront: MFMA busy %7.0f cycles, VALU+MFMA co-executing %5.0f, VALU instructions %6.0f, "
          "lifetime %7.0f cycles" % (k, w, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / w, c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / w,
                                     c.get("SQ_INSTS_VALU", 0) / w, 4 * c.get("SQ_WAVE_CYCLES", 0) / w))
