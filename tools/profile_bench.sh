#!/bin/bash
# rocprofv3 passes of one bench.py command (run on the GPU box through gpurun):
#   tools/profile_bench.sh <tag> <bench.py args...>
# writes gpurun_out/prof_<tag>/{trace,pmc1..pmc4}; tools/prof_summary.py turns them into the
# kernel_stats.csv + pmc_summary.json that get committed under profiles/.
# Counters are collected in passes of their own (no tracing beside --pmc, program directly after --).
set -e -o pipefail
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 bench.py "$@" > "$out/trace.json" 2> "$out/trace.err"
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "WRITE_SIZE GRBM_GUI_ACTIVE" \
  "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out/pmc$i" -o run -- python3 bench.py "$@" > "$out/pmc$i.json" 2> "$out/pmc$i.err"
done
python3 tools/prof_summary.py "$out" > "$out/summary.txt"
cat "$out/summary.txt"
# the raw per-dispatch tables are tens of MB (gpurun copies back at most 64 MiB): keep the summaries only
rm -rf "$out/trace" "$out"/pmc[0-9]
