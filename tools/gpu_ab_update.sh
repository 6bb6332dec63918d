cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
rm -f gpurun_out/ab_update.log
for r in 1 2; do
for f in $V/libirrl_env_*.so; do
  n=$(basename $f .so)
  IRRL_ENV_LIB=$PWD/$f timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n ppo rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms')" >> gpurun_out/ab_update.log
done
done
IRRL_ENV_LIB=$PWD/$V/libirrl_env_occ2.so timeout 900 python -m pytest tests/test_gpu_ppo.py -q -x > gpurun_out/pytest_occ2.log 2>&1
echo done
