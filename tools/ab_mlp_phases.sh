V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
for v in base ab1 ab2 ab3 ab4 ab5 ab6 ab7; do echo $v; IRRL_ENV_LIB=$PWD/$V/libirrl_env_$v.so python tools/mlp_kernel_time.py 2>&1 | grep "bf16x3  indexed"; done
