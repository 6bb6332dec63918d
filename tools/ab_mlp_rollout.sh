for r in 1 2; do
timeout 300 python tools/ppo_bench.py --policy mlp --envs 4096 --iters 5 --cfg bp5_imitation.yaml 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlp rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')"
done
