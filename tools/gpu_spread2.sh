cd $GRAFT_REPO_ROOT
V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants
rm -f gpurun_out/wave_spread2.log
for solver in 2 0; do
IRRL_ENV_LIB=$PWD/$V/libirrl_env_prof.so timeout 300 python tools/wave_spread.py --cfg default_cfg.yaml --sigma 1.0 --solver $solver 2>/dev/null | tail -1 >> gpurun_out/wave_spread2.log
done
echo done
