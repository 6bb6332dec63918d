cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/p; rm -f gpurun_out/p/nodx2.log
L=$PWD/high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants/libirrl_env_nodx2.so
IRRL_ENV_LIB=$L IRRL_LSTM_BWD_SHARE=11 python -m pytest tests/test_gpu_ppo.py -x -q > gpurun_out/p/pytest.log 2>&1; tail -2 gpurun_out/p/pytest.log
for cfg in "base:10:0" "base:10:1" "nodx2:11:0" "nodx2:11:1" "nodx2:10:1" "base:10:0" "base:10:1" "nodx2:11:1"; do
  IFS=: read lib sh rs <<< "$cfg"
  if [ $lib = base ]; then unset IRRL_ENV_LIB; else export IRRL_ENV_LIB=$L; fi
  IRRL_LSTM_ROLE_SHIFT=$rs IRRL_LSTM_BWD_SHARE=$sh python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg update %.2f ms'%(d['update_s']*1e3))" >> gpurun_out/p/nodx2.log
done
cat gpurun_out/p/nodx2.log
