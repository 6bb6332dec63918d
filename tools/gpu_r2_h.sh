cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --cpu-seconds 0 > gpurun_out/bench.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 > gpurun_out/bench_driver_style.log 2>&1
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
IRRL_ENV_LIB=$PWD/$V/libirrl_env_prof.so timeout 300 python tools/wave_spread.py > gpurun_out/wave_spread.log 2>&1
IRRL_ENV_LIB=$PWD/$V/libirrl_env_prof.so timeout 300 python tools/wave_spread.py --cfg default_cfg.yaml >> gpurun_out/wave_spread.log 2>&1
echo done
