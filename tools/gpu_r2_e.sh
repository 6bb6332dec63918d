cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 > gpurun_out/ppo_lstm.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ppo_r2 -- python3 $GRAFT_REPO_ROOT/tools/ppo_bench.py --policy lstm --envs 4096 --iters 2 --epochs 2 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_ppo.log 2>&1
echo done
