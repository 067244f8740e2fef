# 13x13 / 65 536 rollout in fresh processes: where the dispatcher put the workgroups and how fast each XCC ran them
for n in 13 13 13 11; do
  echo "== process N=$n"
  N=$n T=20 HEXVEC_SO=hex_gym_env_amd/csrc/variants/stamps.so timeout -k 10 120 python tools/exp_placement.py 2>&1 | grep -v amdgpu.ids | grep "by XCC\|correlation" | tail -10
done
