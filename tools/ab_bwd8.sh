#!/bin/bash
# same-box A/B of the backward sequence kernel's wave layouts (IRRL_LSTM_BWD8: 0 = four waves, round 4; 1 = 3 main + 5 consumer waves;
# 2 = 6 half-main + 2 weight-gradient waves), interleaved, PPO-LSTM iteration at the default arithmetic
O=gpurun_out
rm -f $O/abbwd8.log
run() { IRRL_LSTM_BWD8=$1 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('IRRL_LSTM_BWD8=$1 rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')" >> $O/abbwd8.log; }
for r in 1 2; do for m in 2 0 1; do run $m; done; done
