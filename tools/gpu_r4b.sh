#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
rm -f $O/lstm_nt_ab.log
for r in 1 2; do for lib in "" $PWD/$V/libirrl_env_lbfnt.so; do
  IRRL_ENV_LIB=$lib timeout 600 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$(basename "$lib")', 'rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')" >> $O/lstm_nt_ab.log 2>&1
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pybind" > $O/pybind_tests.log 2>&1; echo "rc=$?" >> $O/pybind_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --ppo-iters 0 --cpu-seconds 0 --check-steps 0 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('per_step_call', d['per_step_call']); print('per_step_call_compiled', d['per_step_call_compiled'])" >> $O/pybind_tests.log 2>&1
