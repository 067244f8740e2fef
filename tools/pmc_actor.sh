#!/bin/bash
# counter passes on the fused actor kernel alone: tools/pmc_actor.sh  (N / E through the environment)
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/pmc_actor
rm -rf "$out"; mkdir -p "$out"
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out/pmc$i" -o run -- python3 tools/run_actor_only.py > "$out/pmc$i.log" 2> "$out/pmc$i.err" || { tail -3 "$out/pmc$i.err"; continue; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_actor/pmc*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mlp64" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k:34s} {sum(v)/len(v):14.1f}  (mean of {len(v)} launches)")
PY
rm -rf "$out"/pmc[0-9]
