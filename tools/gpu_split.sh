cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
for r in 1 2; do
for f in $V/libirrl_env_*.so; do
  n=$(basename $f .so)
  echo "$n $(IRRL_ENV_LIB=$PWD/$f timeout 300 python tools/step_cost_split.py 2>/dev/null | tail -1)" >> gpurun_out/split.log
done
done
echo done
