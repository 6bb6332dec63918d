cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 > gpurun_out/bench_driver_style.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 --no-graph > gpurun_out/bench_driver_style_nograph.log 2>&1
timeout 600 python bench.py --cpu-seconds 0 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
echo done
