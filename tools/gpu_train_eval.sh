# Stage 1 (imitation, the reference's command line at 3x its sample budget), headless evaluation of the result, then the
# relaxation-stage warm start (--load, half the learning rate) for a short continuation.  Artifacts under gpurun_out/irrl/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/irrl
timeout 1500 python scripts/run_bp_v5.py --train --l 0.001 --max_iter 600000000 --eval_every_n 0 2>&1 | grep -E "nupdates|final checkpoint" | awk 'NR%25==1 || /final/' | cut -c1-330 > gpurun_out/irrl/stage1.log
CK=$(grep "final checkpoint" gpurun_out/irrl/stage1.log | awk '{print $3}')
cp "$CK" gpurun_out/irrl/stage1_final.pkl
for c in 1.0 2.0 3.0; do
  timeout 300 python scripts/run_bp_v5.py --test --cfg high_speed_quadrupedal_locomotion_by_irrl_amd/rsc/bp5_test.yaml --model gpurun_out/irrl/stage1_final.pkl --cmd $c --steps 2000 2>&1 | grep "^test:" >> gpurun_out/irrl/eval.log
done
timeout 600 python scripts/run_bp_v5.py --train --l 0.0005 --max_iter 60000000 --eval_every_n 0 --load gpurun_out/irrl/stage1_final.pkl 2>&1 | grep -E "nupdates" | awk 'NR%40==1' | cut -c1-330 > gpurun_out/irrl/stage2.log
echo done
