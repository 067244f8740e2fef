# parity subset + host-stepped A/B of qpb variants: bash tools/run_ab_step.sh name...
set -e
V=hex_gym_env_amd/csrc/variants
for v in "$@"; do
  HEXVEC_SO=$PWD/$V/$v.so python -m pytest tests/test_gpu_parity.py -x -q -k "bitboard and (selfplay_rng_steps or soak_all_sizes or external_actions or two_sided_steps) and 11" > gpurun_out/r3b/parity_$v.log 2>&1 || { echo "PARITY FAIL $v"; tail -20 gpurun_out/r3b/parity_$v.log; exit 1; }
  echo "parity ok $v: $(tail -1 gpurun_out/r3b/parity_$v.log)"
done
args=""
for v in "$@"; do args="$args $V/$v.so"; done
python tools/ab_step.py $args 2>&1 | grep -v amdgpu.ids
