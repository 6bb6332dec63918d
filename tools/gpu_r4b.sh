#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/lstm_prec_ppo.log
for p in f32 bf16x3 bf16x6 f32 bf16x3 bf16x6; do
  IRRL_LSTM_PRECISION=$p timeout 600 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 5 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$p', 'rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s', d['iters_per_sec_min_median_max'])" >> $O/lstm_prec_ppo.log 2>&1
done
