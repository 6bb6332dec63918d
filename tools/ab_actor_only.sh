V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
O=gpurun_out
rm -f $O/abactor.log
for r in 1 2; do
  for spec in "actorcarry 3" "actornocarry 3" "actorcarry 2"; do
    set -- $spec
    IRRL_ENV_LIB=$PWD/$V/libirrl_env_$1.so IRRL_ROLLOUT_FUSED=$2 timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 IRRL_ROLLOUT_FUSED=$2 rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')" >> $O/abactor.log
  done
done
