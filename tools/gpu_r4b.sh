#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
rm -f $O/lstm_depth_ab.log
for r in 1 2; do for lib in "" $PWD/$V/libirrl_env_d2.so $PWD/$V/libirrl_env_d4.so; do
  IRRL_ENV_LIB=$lib timeout 600 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$(basename "$lib")', 'rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')" >> $O/lstm_depth_ab.log 2>&1
done; done
