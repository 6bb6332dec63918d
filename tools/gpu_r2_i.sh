cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --cpu-seconds 0 > gpurun_out/bench.log 2>&1
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
rm -f gpurun_out/wave_spread.log
IRRL_ENV_LIB=$PWD/$V/libirrl_env_prof.so timeout 300 python tools/wave_spread.py 2>/dev/null | tail -1 >> gpurun_out/wave_spread.log
IRRL_ENV_LIB=$PWD/$V/libirrl_env_prof.so timeout 300 python tools/wave_spread.py --cfg default_cfg.yaml --sigma 1.0 2>/dev/null | tail -1 >> gpurun_out/wave_spread.log
echo done
