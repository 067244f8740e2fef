set -e
V=hex_gym_env_amd/csrc/variants
mkdir -p gpurun_out/r3b
for v in "$@"; do
  HEXVEC_SO=$PWD/$V/$v.so python -m pytest tests/test_gpu_parity.py -x -q -k "rollout1_at_bench or rollout_selfplay_vs or episode_sums" > gpurun_out/r3b/parity_$v.log 2>&1 || { echo "PARITY FAIL $v"; tail -20 gpurun_out/r3b/parity_$v.log; exit 1; }
  echo "parity ok $v: $(tail -1 gpurun_out/r3b/parity_$v.log)"
done
args=""
for v in "$@"; do args="$args $V/$v.so"; done
T=20 python tools/ab_rollout.py $args
T=200 python tools/ab_rollout.py $args
