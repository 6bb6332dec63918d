# same-box A/B of the actor-only LSTM rollout: the actor as each wave's own work (default) against the workgroup-wide actor step (IRRL_ACTOR_WAVES=0)
for r in 1 2; do for w in 0 1; do
IRRL_ACTOR_WAVES=$w timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('IRRL_ACTOR_WAVES=$w rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s', round(d['env_steps_per_sec_in_rollout']/1e6,1), 'M env-steps/s in the rollout')"
done; done
