cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/o; rm -f gpurun_out/o/pipe.log
IRRL_LSTM_BWD_PIPE=1 python -m pytest tests/test_gpu_ppo.py -x -q > gpurun_out/o/pytest_pipe.log 2>&1; tail -2 gpurun_out/o/pytest_pipe.log
for v in 0 1 0 1; do
IRRL_LSTM_BWD_PIPE=$v python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipe $v update %.2f ms'%(d['update_s']*1e3))" >> gpurun_out/o/pipe.log
done
cat gpurun_out/o/pipe.log
