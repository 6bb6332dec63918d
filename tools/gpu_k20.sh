# the driver's command line (--steps 20 --warmup 5): how much of the bracket is launch / synchronisation latency?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/k20.log
one() { # label, env assignments..., then bench args after --
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for r in 1 2 3; do
    env "${envs[@]}" python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 --check-steps 0 "$@" 2>/dev/null | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step']*1e3,2), 'us/step wall,', round(d['roofline']['avg_launch_us'],2), 'us/step events,', round(d['value']/1e6,1), 'M')" >> gpurun_out/k20.log
  done
}
one rows X=1 -- --launch rows
one graph X=1 -- --launch graph
one python X=1 -- --launch python
one rows_nointerrupt HSA_ENABLE_INTERRUPT=0 -- --launch rows
one graph_nointerrupt HSA_ENABLE_INTERRUPT=0 -- --launch graph
one rows_devkernarg HIP_FORCE_DEV_KERNARG=1 -- --launch rows
one rows_2000 X=1 -- --launch rows --steps 2000
cat gpurun_out/k20.log
