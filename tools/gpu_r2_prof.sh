cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_* gpurun_out/prof_r2_bench
bash tools/gpu_pmc.sh
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 --check-steps 500 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 > gpurun_out/bench_driver_style.log 2>&1
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
echo done
