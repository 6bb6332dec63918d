cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
timeout 600 python bench.py --cfg bp5_terrain.yaml --cpu-seconds 0 --ppo-iters 0 --steps 2000 > gpurun_out/bench_terrain.log 2>&1
timeout 600 python bench.py --cfg default_cfg.yaml --cpu-seconds 0 --ppo-iters 0 --steps 2000 > gpurun_out/bench_train.log 2>&1
bash tools/gpu_sweep.sh
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_l16 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --cpu-seconds 0 --ppo-iters 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
echo done
