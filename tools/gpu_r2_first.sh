cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > gpurun_out/box.log 2>&1
timeout 120 tools/microbench/valu_issue > gpurun_out/valu_issue.log 2>&1
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 > gpurun_out/bench_driver_style.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 --check-steps 500 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
echo done
