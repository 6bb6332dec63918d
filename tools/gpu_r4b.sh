#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/mlp_grad_error.py 768000 > $O/mlp_grad_error.log 2>&1
timeout 900 python -m pytest tests/test_gpu_ppo.py -m gpu -q -x -k "mlp_policy_gradient or mlp_ppo_update or flat_optimizer" > $O/mlp_bf16_tests.log 2>&1; echo "rc=$?" >> $O/mlp_bf16_tests.log
for p in f32 bf16x3 f32 bf16x3; do
  IRRL_MLP_PRECISION=$p timeout 600 python tools/ppo_bench.py --policy mlp --envs 4096 --iters 5 --cfg bp5_imitation.yaml 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$p', 'rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')" >> $O/mlp_grad_error.log 2>&1
done
