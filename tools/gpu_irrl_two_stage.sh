# The IRRL recipe on one MI355X at the benchmark scale: stage 1 "imitation" (rsc/bp5_imitation.yaml: joint-imitation-heavy
# reward, no noise / randomisation), stage 2 "relaxation" (--load, half the learning rate, rsc/default_cfg.yaml: relaxed
# imitation, velocity / torque terms up, observation noise + randomised dynamics).  Logs under gpurun_out/irrl2/.
cd $GRAFT_REPO_ROOT
R=high_speed_quadrupedal_locomotion_by_irrl_amd/rsc
mkdir -p gpurun_out/irrl2; rm -f gpurun_out/irrl2/*
timeout 900 python scripts/run_bp_v5.py --train --cfg $R/bp5_imitation.yaml --num_envs 4096 --l 0.001 --max_iter $((4096*750*300)) --eval_every_n 0 2>&1 | grep -E "nupdates|final checkpoint" | awk 'NR%10==1 || /final/' | cut -c1-330 > gpurun_out/irrl2/stage1_imitation.log
CK=$(grep "final checkpoint" gpurun_out/irrl2/stage1_imitation.log | awk '{print $3}')
cp "$CK" gpurun_out/irrl2/stage1.pkl
timeout 900 python scripts/run_bp_v5.py --train --cfg $R/default_cfg.yaml --num_envs 4096 --l 0.0005 --max_iter $((4096*750*300)) --eval_every_n 0 --load gpurun_out/irrl2/stage1.pkl 2>&1 | grep -E "nupdates|final checkpoint" | awk 'NR%10==1 || /final/' | cut -c1-330 > gpurun_out/irrl2/stage2_relaxation.log
CK2=$(grep "final checkpoint" gpurun_out/irrl2/stage2_relaxation.log | awk '{print $3}')
cp "$CK2" gpurun_out/irrl2/stage2.pkl
timeout 300 python tools/eval_checkpoint_gpu.py gpurun_out/irrl2/stage2.pkl 256 2>&1 | grep -E "rollout|deterministic" > gpurun_out/irrl2/eval_stage2.log
echo done
