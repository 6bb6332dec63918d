cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ppo.py -x -q -s > gpurun_out/pytest_gpu_ppo.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu_ppo.log
timeout 900 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 --epochs 10 > gpurun_out/ppo_lstm.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ppo -- python3 $GRAFT_REPO_ROOT/tools/ppo_bench.py --policy lstm --envs 4096 --iters 2 --epochs 2 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_ppo.log 2>&1
echo done
