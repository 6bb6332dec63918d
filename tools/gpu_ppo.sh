cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ppo.py -x -q -s > gpurun_out/pytest_gpu_ppo.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu_ppo.log
timeout 600 python tools/ppo_bench.py --policy mlp --envs 4096 --iters 3 > gpurun_out/ppo_mlp.log 2>&1
timeout 1500 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 2 --epochs 2 > gpurun_out/ppo_lstm.log 2>&1
echo done
