# in-process A/B where the variants are the SAME library under different environment settings is not possible (the
# setting is read once per process): run the processes back to back on several buffer sets instead
V=hex_gym_env_amd/csrc/variants
for t in 20 100 200; do
  for spec in 0 1; do
    echo "T=$t HEXVEC_LPB_SPEC=$spec"; HEXVEC_LPB_SPEC=$spec T=$t SETS=4 python tools/ab_inproc.py $V/base.so $V/prod3.so 2>&1 | grep -v amdgpu | grep "round 1"
  done
done
