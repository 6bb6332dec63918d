V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
IRRL_ENV_LIB=$PWD/$V/libirrl_env_mbprof.so python tools/mlp_bf16_phases.py 2>&1 | grep kind
for r in 1 2; do for v in pace0 pace1 pace2 pace3; do echo $v; IRRL_ENV_LIB=$PWD/$V/libirrl_env_$v.so python tools/mlp_kernel_time.py 2>&1 | grep "bf16x3  indexed"; done; done
