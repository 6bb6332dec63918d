#!/bin/bash
# round-4 one-off: driver-style bracket repeated, ContactExit A/B on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
line() { python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{') and 'metric' in l:
        d=json.loads(l); print('$1', round(d['value']/1e6,2), 'M env-steps/s', 'ms_per_step', round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['avg_launch_us'],2), 'us', d['config'].get('launch_probe_us_per_step'), d['config']['launch'][:12])"; }
rm -f $O/r4a.log
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --ppo-iters 0 --cpu-seconds 0 --check-steps 0 2>/dev/null | line "driver-style auto $i" >> $O/r4a.log
done
for m in rows graph python; do
  timeout 300 python bench.py --steps 20 --warmup 5 --ppo-iters 0 --cpu-seconds 0 --check-steps 0 --launch $m 2>/dev/null | line "driver-style $m" >> $O/r4a.log
done
for r in 1 2; do for e in 0 1; do
  timeout 300 python bench.py --set ContactExit=$e --cpu-seconds 0 --ppo-iters 0 --steps 2000 --check-steps 0 2>/dev/null | line "ContactExit=$e" >> $O/r4a.log
  timeout 300 python bench.py --cfg default_cfg.yaml --set ContactExit=$e --cpu-seconds 0 --ppo-iters 0 --steps 2000 --check-steps 0 2>/dev/null | line "default_cfg ContactExit=$e" >> $O/r4a.log
done; done
timeout 300 python bench.py --cfg bp5_terrain.yaml --cpu-seconds 0 --ppo-iters 0 --steps 2000 --check-steps 0 2>/dev/null | line "terrain" >> $O/r4a.log
