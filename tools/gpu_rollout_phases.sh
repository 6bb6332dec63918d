cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=$PWD/high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants/libirrl_env_pol.so
echo "== fused" > gpurun_out/rollout_phases.log
IRRL_ENV_LIB=$L IRRL_ROLLOUT_FUSED=1 python tools/rollout_phases.py 2>/dev/null >> gpurun_out/rollout_phases.log
echo "== two launches" >> gpurun_out/rollout_phases.log
IRRL_ENV_LIB=$L python tools/rollout_phases.py 2>/dev/null >> gpurun_out/rollout_phases.log
cat gpurun_out/rollout_phases.log
