cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q
IRRL_ENV_LIB=$PWD/high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants/libirrl_env_fwd.so python tools/lstm_fwd_phases.py 2>/dev/null > gpurun_out/q/fwd_phases.log
python -m pytest tests/test_gpu_ppo.py -x -q > gpurun_out/q/pytest.log 2>&1; tail -2 gpurun_out/q/pytest.log
for v in 1 2 3; do
python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('update %.2f ms'%(d['update_s']*1e3))"
done
cat gpurun_out/q/fwd_phases.log
