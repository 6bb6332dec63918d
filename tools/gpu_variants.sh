# one-call A/B of the variant libraries built by tools/build_variants.py (same box, interleaved, 2 rounds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
for r in 1 2; do
for f in $V/libirrl_env_*.so; do
  n=$(basename $f .so)
  for lanes in 16 4; do
    envs=4096; [ $lanes = 4 ] && envs=32768
    IRRL_ENV_LIB=$PWD/$f IRRL_LANES_PER_ROBOT=$lanes timeout 300 python bench.py --cpu-seconds 0 --ppo-iters 0 --steps 2000 --envs $envs 2>/dev/null | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n L$lanes', round(d['value']/1e6,2), 'M env-steps/s', round(d['roofline']['avg_launch_us'],2), 'us')" >> gpurun_out/variants.log
  done
done
done
for f in $V/libirrl_env_*.so; do
  n=$(basename $f .so)
  IRRL_ENV_LIB=$PWD/$f timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n ppo rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms')" >> gpurun_out/variants.log
  IRRL_ENV_LIB=$PWD/$f timeout 300 python tools/step_cost_split.py 2>/dev/null | tail -1 >> gpurun_out/variants.log
done
echo done
