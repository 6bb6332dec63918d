# same-box interleaved A/B of the variant libraries (16-lane layout, 4096 envs, 3 rounds), plus the GPU test given as $1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/variants.log
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
for r in 1 2 3; do
for f in $V/libirrl_env_*.so; do
  n=$(basename $f .so)
  IRRL_ENV_LIB=$PWD/$f timeout 300 python bench.py --cpu-seconds 0 --ppo-iters 0 --steps 2000 2>/dev/null | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['value']/1e6,2), 'M env-steps/s', round(d['roofline']['avg_launch_us'],2), 'us')" >> gpurun_out/variants.log
done
done
if [ -n "$1" ]; then python -m pytest tests/test_gpu_parity.py -x -q -k "$1" -s > gpurun_out/ab_pytest.log 2>&1; tail -3 gpurun_out/ab_pytest.log; fi
cat gpurun_out/variants.log
