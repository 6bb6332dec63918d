cd $GRAFT_REPO_ROOT
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
rm -f gpurun_out/wave_spread3.log
IRRL_ENV_LIB=$PWD/$V/libirrl_env_prof.so timeout 300 python tools/wave_spread.py 2>/dev/null | tail -1 >> gpurun_out/wave_spread3.log
IRRL_ENV_LIB=$PWD/$V/libirrl_env_prof.so timeout 300 python tools/wave_spread.py --cfg default_cfg.yaml --sigma 1.0 2>/dev/null | tail -1 >> gpurun_out/wave_spread3.log
cat gpurun_out/wave_spread3.log
