# timing-only A/B (variants whose results are deliberately wrong): bash tools/run_ab_noparity.sh T name...
set -e
V=hex_gym_env_amd/csrc/variants
T=$1; shift
args=""
for v in "$@"; do args="$args $V/$v.so"; done
T=$T python tools/ab_rollout.py $args 2>&1 | grep -v amdgpu.ids
