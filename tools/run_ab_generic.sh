# bash tools/run_ab_generic.sh "<env assignments>" name...   (parity subset first, then tools/ab_rollout.py)
set -e
V=hex_gym_env_amd/csrc/variants
envs=$1; shift
mkdir -p gpurun_out/ab
for v in "$@"; do
  HEXVEC_SO=$PWD/$V/$v.so python -m pytest tests/test_gpu_parity.py -x -q -k "rollout1_at_bench or rollout_selfplay_vs" > gpurun_out/ab/parity_$v.log 2>&1 || { echo "PARITY FAIL $v"; tail -20 gpurun_out/ab/parity_$v.log; exit 1; }
  echo "parity ok $v: $(tail -1 gpurun_out/ab/parity_$v.log)"
done
args=""
for v in "$@"; do args="$args $V/$v.so"; done
env $envs python tools/ab_rollout.py $args 2>&1 | grep -v amdgpu.ids
