# same-box A/B: both layers of a stack in one forward launch (default) against one launch per layer (IRRL_LSTM_FUSE_STACK=0)
for r in 1 2 3; do for f in 0 1; do for p in bf16x3; do
IRRL_LSTM_FUSE_STACK=$f timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 --precision $p 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('IRRL_LSTM_FUSE_STACK=$f $p rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')"
done; done; done
