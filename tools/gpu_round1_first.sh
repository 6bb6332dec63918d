cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > gpurun_out/box.log 2>&1
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2000 --warmup 100 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --cpu-seconds 0 --ppo-iters 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
echo done
