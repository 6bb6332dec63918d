# Stage 1 (imitation, the reference's command line: 200 envs, 2e8 samples), headless evaluation of the result, then the
# relaxation-stage warm start (--load, half the learning rate) for a short continuation.  Artifacts under gpurun_out/irrl/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/irrl
rm -f gpurun_out/irrl/*.log
timeout 1500 python scripts/run_bp_v5.py --train --l 0.001 --max_iter ${MAX_ITER:-200000000} --eval_every_n 0 2>&1 | grep -E "nupdates|final checkpoint" | awk 'NR%25==1 || /final/' | cut -c1-330 > gpurun_out/irrl/stage1.log
CK=$(grep "final checkpoint" gpurun_out/irrl/stage1.log | awk '{print $3}')
cp "$CK" gpurun_out/irrl/stage1_final.pkl
timeout 300 python tools/eval_checkpoint_gpu.py gpurun_out/irrl/stage1_final.pkl 256 2>&1 | grep -E "rollout|deterministic" >> gpurun_out/irrl/eval.log
for c in 1.0 2.0 3.0; do
  timeout 300 python scripts/run_bp_v5.py --test --model gpurun_out/irrl/stage1_final.pkl --cmd $c --steps 2000 2>&1 | grep "^test:" >> gpurun_out/irrl/eval.log
done
timeout 600 python scripts/run_bp_v5.py --train --l 0.0005 --max_iter 30000000 --eval_every_n 0 --save 0 --load gpurun_out/irrl/stage1_final.pkl 2>&1 | grep -E "nupdates" | awk 'NR%20==1' | cut -c1-330 > gpurun_out/irrl/stage2.log
echo done
