V=hex_gym_env_amd/csrc/variants
for cfg in "7 32768" "13 16384" "11 65536"; do set -- $cfg; echo "N=$1 E=$2 (bitboard rollout)"; N=$1 E=$2 T=100 SETS=2 LAYOUT=bitboard python tools/ab_inproc.py $V/qb0.so $V/qb1.so 2>&1 | grep -v amdgpu | grep "round 1"; done
