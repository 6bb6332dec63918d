#!/bin/bash
# ONE parameterised GPU-box script (replaces the tools/gpu_*.sh pile of rounds 1-2).  Run through gpurun, e.g.
#   gpurun --timeout 2400 -- 'bash tools/gpu.sh box smoke tests bench'
# Every stage writes under gpurun_out/ (merged back by gpurun); copy what is to be judged into profiles/ by hand.
# Stages:
#   box            GPU / host identification
#   smoke          __graft_entry__.smoke()
#   tests [expr]   pytest -m gpu (optionally -k expr via TESTS_K=...)
#   bench          bench.py default line + driver-style 20-step line
#   bench3         the three shipped configs, env leg only
#   solvers        env leg of the benchmark config under ContactSolver 0 1 2 3
#   sweep          envs-per-GPU sweep + PCIe-inclusive numpy boundary
#   gloo8          bench.py --gpus 8 with all eight ranks on this one device (gloo): configs 4 / 5 at 32 768 envs, PPO collectives in the loop
#   prof           rocprofv3 --kernel-trace --stats of the bench command
#   profppo        rocprofv3 --kernel-trace --stats of 2 PPO iterations x 2 epochs (LSTM)
#   profmlp        rocprofv3 --kernel-trace --stats of 3 PPO iterations with the MlpPolicy learner (config 2)
#   pmc            PMC passes of the env step kernel (separate --pmc runs, no trace domains besides kernel-trace)
#   pmcsum         tools/pmc_summarize.py on the box (name: PMC_NAME), copies of the two JSON files into gpurun_out/
#   pmclstm        PMC passes of the LSTM sequence kernels
#   pmcmlp         PMC passes of the MlpPolicy gradient kernels
#   ppo            tools/ppo_bench.py lstm + mlp, 3 iterations each
#   irrl2          the IRRL recipe at the benchmark scale: stage 1 imitation + stage 2 relaxation, 4096 envs, 300 updates each
#   trainmlp       BASELINE config 2 as a training run: MlpPolicy on bp5_imitation.yaml, 4096 envs, 400 updates through the gradient kernels + evaluation
#   terrain        BASELINE config 5 on one GPU: 4096 envs on the Perlin height field with per-episode friction / mass / COM randomisation and the
#                  command process, LSTM policy, 200 updates from scratch + evaluation of the result
#   train200       the reference's command line (200 envs, 2e8 samples), headless evaluation of the result
#   abprec         config 3 trained 300 updates per (seed, arm): arms = the LSTM update's arithmetic (bf16x3, bf16x6, f32; ABPREC_ARMS can set the
#                  actor's and the critic's stacks apart), ABPREC_SEEDS seeds each (default 1 2 3); tools/ab_precision_table.py tabulates mean +- spread
#   variants       A/B of every csrc/_variants/libirrl_env_*.so (tools/build_variants.py) on this one box, interleaved
#   ablstm         same-box A/B of the PPO-LSTM update (bf16x3 and bf16x6) over every csrc/_variants/libirrl_env_*.so
#   abmlp          same-box A/B of the PPO-MLP update with / without the packed sample records (IRRL_MLP_RECORDS)
#   abmlpw         same-box A/B of the MlpPolicy gradient kernels, four waves against producer / consumer wave pairs (IRRL_MLP_WAVES)
#   abrecomp       same-box A/B of the PPO-LSTM update (bf16x3) with the recomputing backward kernel on / off (IRRL_LSTM_RECOMPUTE)
#   graderr        LSTM sequence kernels vs float64 autograd at 750 x 4096 (every arithmetic), shipped library + every variant library
#   spread         per-wave durations of the step kernel (needs the `prof` variant library)
#   ab <cmd...>    run the rest of the line verbatim (one-off A/B)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p "$O"
line() { python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{') and 'metric' in l:
        d=json.loads(l); print('$1', round(d['value']/1e6,2), 'M env-steps/s', round(d['roofline']['avg_step_us'],2), 'us per step', 'fp32_frac', round(d['roofline']['frac'],4), *(('persistent', round(d['launch_modes']['persistent']['us_per_step'],2), 'per-step launch', round(d['launch_modes']['rows']['us_per_step'],2)) if 'rows' in (d.get('launch_modes') or {}) else ()))"; }
while [ $# -gt 0 ]; do
  stage=$1; shift
  case $stage in
    box) (rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name"; python3 -c "import os;print('cgroup cpus', len(os.sched_getaffinity(0)))") > $O/box.log 2>&1 ;;
    smoke) timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log ;;
    tests) timeout 2400 python -m pytest tests -m gpu -q -s ${TESTS_K:+-k "$TESTS_K"} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ;;
    bench)
      timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
      timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.log 2>&1 ;;
    bench3)
      for c in bp5_imitation default_cfg bp5_terrain; do
        timeout 600 python bench.py --cfg $c.yaml --cpu-seconds 0 --ppo-iters 0 --steps 2000 > $O/bench_$c.log 2>&1
      done ;;
    solvers)
      rm -f $O/solvers.log
      for s in 3 2 1 0 3 2; do
        timeout 600 python bench.py --set ContactSolver=$s --cpu-seconds 0 --ppo-iters 0 --steps 2000 2>/dev/null | line "ContactSolver=$s" >> $O/solvers.log
      done ;;
    sweep)
      rm -f $O/sweep.log
      for n in 1024 4096 6144 8192 16384 32768 131072; do
        timeout 300 python bench.py --cpu-seconds 0 --ppo-iters 0 --steps 1000 --warmup 100 --check-steps 0 --envs $n --launch rows 2>/dev/null | line "envs=$n, one launch per step:" >> $O/sweep.log
        timeout 300 python bench.py --cpu-seconds 0 --ppo-iters 0 --steps 1000 --warmup 100 --check-steps 0 --envs $n --launch persistent 2>/dev/null | line "envs=$n, ONE persistent launch:" >> $O/sweep.log
      done
      timeout 300 python tools/host_boundary_rate.py >> $O/sweep.log 2>&1 ;;
    gloo8)
      # BASELINE configs 4 / 5 at their real size WITHOUT the 8-GPU node: eight ranks x 4096 envs = the 32 768-env job, every rank on
      # this one MI355X, barrier / all-reduces over gloo (the RCCL / xGMI transport itself stays unmeasured); the PPO legs run the
      # per-optimizer-step collectives of SURVEY 8e.  The JSON line is copied to profiles/ by hand.
      IRRL_BENCH_BACKEND=gloo IRRL_BENCH_ONE_DEVICE=1 OMP_NUM_THREADS=2 timeout 1500 python bench.py --gpus 8 --ppo-iters ${GLOO8_PPO_ITERS:-2} > $O/bench_gloo8.log 2>&1; echo "rc=$?" >> $O/bench_gloo8.log
      IRRL_BENCH_BACKEND=gloo IRRL_BENCH_ONE_DEVICE=1 OMP_NUM_THREADS=2 timeout 900 python bench.py --gpus 8 --cfg bp5_terrain.yaml --ppo-iters 0 > $O/bench_gloo8_terrain.log 2>&1; echo "rc=$?" >> $O/bench_gloo8_terrain.log ;;
    prof)
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 --check-steps 0 > $O/rocprof_bench.log 2>&1) ;;
    profppo)
      (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ppo -- python3 $R/tools/ppo_bench.py --policy lstm --envs 4096 --iters 2 --epochs 2 > $O/rocprof_ppo.log 2>&1)
      # the row-sum kernel's trace durations against what ran beside it (verdict r4 item 11)
      python3 tools/kernel_overlap.py "$(ls -t $O/prof_ppo/*/*_kernel_trace.csv | head -1)" irrl_sum_rows > $O/sum_rows_overlap.log 2>&1 ;;
    profmlp)
      rm -rf $O/prof_mlp
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mlp -- python3 $R/tools/ppo_bench.py --policy mlp --envs 4096 --iters 3 > $O/rocprof_mlp.log 2>&1) ;;
    pmc)
      rm -rf $O/pmc_env_*
      (cd /tmp && export TMPDIR=/tmp
       for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
         tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
         timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_env_$tag -- python3 $R/tools/pmc_workload.py 300 4096 > $O/pmc_env_$tag.log 2>&1
       done) ;;      # then, back in the build container: python tools/pmc_summarize.py r03_pmc_summary
    pmcsum)   # behind `pmc`, in front of `bench` in the same call: the summary (with this library's hash) is there when bench.py looks for `traffic`
      python3 tools/pmc_summarize.py ${PMC_NAME:-pmc_summary} > $O/pmcsum.log 2>&1; cp profiles/${PMC_NAME:-pmc_summary}.json profiles/pmc_summary_latest.json $O/ 2>/dev/null ;;
    pmclstm)
      rm -rf $O/pmc_lstm_*
      (cd /tmp && export TMPDIR=/tmp
       for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
         tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
         timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_lstm_$tag -- python3 $R/tools/ppo_bench.py --policy lstm --envs 4096 --iters 1 --epochs 1 > $O/pmc_lstm_$tag.log 2>&1
       done) ;;
    pmcmlp)
      rm -rf $O/pmc_mlp_*
      (cd /tmp && export TMPDIR=/tmp
       for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
         tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
         timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_mlp_$tag -- python3 $R/tools/pmc_mlp_workload.py > $O/pmc_mlp_$tag.log 2>&1
       done) ;;      # then: python tools/pmc_summarize_lstm.py r05_pmc_mlp_kernels mlp   (both ways of reading the samples: <kind, false> arrays, <kind, true> packed records)
    ppo)
      timeout 300 python tools/ppo_bench.py --policy mlp --envs 4096 --iters 3 > $O/ppo_mlp.log 2>&1
      timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 > $O/ppo_lstm.log 2>&1 ;;
    irrl2)
      RS=high_speed_quadrupedal_locomotion_by_irrl_amd/rsc; mkdir -p $O/irrl2; rm -f $O/irrl2/*
      timeout 900 python scripts/run_bp_v5.py --train --cfg $RS/bp5_imitation.yaml --num_envs 4096 --l 0.001 --max_iter $((4096*750*300)) --eval_every_n 0 2>&1 | grep -E "nupdates|final checkpoint" | awk 'NR%10==1 || /final/' | cut -c1-330 > $O/irrl2/stage1_imitation.log
      cp "$(grep "final checkpoint" $O/irrl2/stage1_imitation.log | awk '{print $3}')" $O/irrl2/stage1.pkl
      timeout 900 python scripts/run_bp_v5.py --train --cfg $RS/default_cfg.yaml --num_envs 4096 --l 0.0005 --max_iter $((4096*750*300)) --eval_every_n 0 --load $O/irrl2/stage1.pkl 2>&1 | grep -E "nupdates|final checkpoint" | awk 'NR%10==1 || /final/' | cut -c1-330 > $O/irrl2/stage2_relaxation.log
      cp "$(grep "final checkpoint" $O/irrl2/stage2_relaxation.log | awk '{print $3}')" $O/irrl2/stage2.pkl
      timeout 300 python tools/eval_checkpoint_gpu.py $O/irrl2/stage2.pkl 256 2>&1 | grep -E "rollout|deterministic" > $O/irrl2/eval_stage2.log ;;
    trainmlp)
      RS=high_speed_quadrupedal_locomotion_by_irrl_amd/rsc; mkdir -p $O/trainmlp; rm -f $O/trainmlp/*
      timeout 900 python scripts/run_bp_v5.py --train --policy mlp --cfg $RS/bp5_imitation.yaml --num_envs 4096 --l ${MLP_LR:-0.0003} --max_iter $((4096*750*${MLP_UPDATES:-400})) --eval_every_n 0 2>&1 | grep -E "nupdates|final checkpoint" | awk 'NR%10==1 || /final/' | cut -c1-330 > $O/trainmlp/train.log
      cp "$(grep "final checkpoint" $O/trainmlp/train.log | awk '{print $3}')" $O/trainmlp/final.pkl
      timeout 300 python tools/eval_checkpoint_gpu.py $O/trainmlp/final.pkl 256 $RS/bp5_imitation.yaml 2>&1 | grep -E "rollout|deterministic" > $O/trainmlp/eval.log ;;
    terrain)
      RS=high_speed_quadrupedal_locomotion_by_irrl_amd/rsc; mkdir -p $O/terrain; rm -f $O/terrain/*
      timeout 900 python scripts/run_bp_v5.py --train --cfg $RS/bp5_terrain.yaml --num_envs 4096 --l 0.001 --max_iter $((4096*750*200)) --eval_every_n 0 2>&1 | grep -E "nupdates|final checkpoint" | awk 'NR%10==1 || /final/' | cut -c1-330 > $O/terrain/train.log
      cp "$(grep "final checkpoint" $O/terrain/train.log | awk '{print $3}')" $O/terrain/final.pkl
      timeout 300 python tools/eval_checkpoint_gpu.py $O/terrain/final.pkl 256 $RS/bp5_terrain.yaml 2>&1 | grep -E "rollout|deterministic" > $O/terrain/eval.log ;;
    train200)
      mkdir -p $O/irrl; rm -f $O/irrl/*.log
      timeout 1500 python scripts/run_bp_v5.py --train --l 0.001 --max_iter ${MAX_ITER:-200000000} --eval_every_n 0 2>&1 | grep -E "nupdates|final checkpoint" | awk 'NR%25==1 || /final/' | cut -c1-330 > $O/irrl/stage1.log
      cp "$(grep "final checkpoint" $O/irrl/stage1.log | awk '{print $3}')" $O/irrl/stage1_final.pkl
      timeout 300 python tools/eval_checkpoint_gpu.py $O/irrl/stage1_final.pkl 256 2>&1 | grep -E "rollout|deterministic" >> $O/irrl/eval.log
      for c in 1.0 2.0 3.0; do
        timeout 300 python scripts/run_bp_v5.py --test --model $O/irrl/stage1_final.pkl --cmd $c --steps 2000 2>&1 | grep "^test:" >> $O/irrl/eval.log
      done ;;
    abprec)
      # verdict r5 item 1: >= 3 seeds per arm.  Arms: the update's arithmetic (IRRL_LSTM_PRECISION) and, to bisect, actor / critic stacks apart
      # (IRRL_LSTM_PRECISION_PI / _V).  ABPREC_ARMS="name:pi:v ...", ABPREC_SEEDS="1 2 3"; every update's reward / explained variance kept.
      RS=high_speed_quadrupedal_locomotion_by_irrl_amd/rsc; mkdir -p $O/abprec; rm -f $O/abprec/*
      for seed in ${ABPREC_SEEDS:-1 2 3}; do for arm in ${ABPREC_ARMS:-bf16x3:bf16x3:bf16x3 bf16x6:bf16x6:bf16x6 f32:f32:f32}; do
        name=${arm%%:*}; rest=${arm#*:}; ppi=${rest%%:*}; pv=${rest#*:}
        IRRL_LSTM_PRECISION_PI=$ppi IRRL_LSTM_PRECISION_V=$pv timeout 900 python scripts/run_bp_v5.py --train --save 0 --seed $seed --cfg $RS/default_cfg.yaml --num_envs 4096 --l 0.001 --max_iter $((4096*750*${ABPREC_UPDATES:-300})) --eval_every_n 0 2>&1 > $O/abprec/raw.log
        grep -E "nupdates" $O/abprec/raw.log | cut -c1-400 > $O/abprec/train_${name}_s$seed.log; tail -3 $O/abprec/raw.log > $O/abprec/tail_${name}_s$seed.log; rm -f $O/abprec/raw.log
      done; done ;;
    variants)
      V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants; rm -f $O/variants.log
      for r in 1 2; do for f in $V/libirrl_env_*.so; do
        IRRL_ENV_LIB=$PWD/$f timeout 300 python bench.py ${VARIANT_ARGS} --cpu-seconds 0 --ppo-iters 0 --steps 2000 2>/dev/null | line "$(basename $f .so)" >> $O/variants.log
      done; done
      if [ -n "$VARIANTS_PPO" ]; then for f in $V/libirrl_env_*.so; do
        for pol in ${VARIANTS_PPO}; do
          IRRL_ENV_LIB=$PWD/$f timeout 300 python tools/ppo_bench.py --policy $pol --envs 4096 --iters 4 --cfg $([ $pol = mlp ] && echo bp5_imitation.yaml || echo default_cfg.yaml) 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $f .so) ppo $pol rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms')" >> $O/variants.log
        done
      done; fi ;;
    ablstm)
      # same-box A/B of the LSTM update over every csrc/_variants/libirrl_env_*.so, interleaved, two rounds: PPO-LSTM iteration with the update at
      # bf16x3 (default) and bf16x6 (f32 level)
      V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants; rm -f $O/ablstm.log
      for r in 1 2; do for f in $V/libirrl_env_*.so; do for prec in bf16x3 bf16x6; do
        IRRL_ENV_LIB=$PWD/$f timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 --precision $prec 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $f .so) $prec rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')" >> $O/ablstm.log
      done; done; done ;;
    abrecomp)
      # same-box A/B of the PPO-LSTM update at bf16x3: the backward kernel recomputes the gates (IRRL_LSTM_RECOMPUTE=1, default) / loads stored ones (0)
      rm -f $O/abrecomp.log
      for r in 1 2 3; do for rc in 1 0; do
        IRRL_LSTM_RECOMPUTE=$rc timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 --precision bf16x3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('IRRL_LSTM_RECOMPUTE=$rc rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')" >> $O/abrecomp.log
      done; done ;;
    graderr)
      # error of the LSTM sequence kernels against float64 autograd at the training shape (tools/lstm_grad_error.py 750 4096 48), for the shipped
      # library and every csrc/_variants/libirrl_env_*.so
      V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants; rm -f $O/graderr.log
      echo "== shipped library" >> $O/graderr.log; timeout 600 python tools/lstm_grad_error.py 750 4096 48 2>/dev/null >> $O/graderr.log
      for f in $V/libirrl_env_*.so; do
        [ -e "$f" ] || continue
        echo "== $(basename $f .so)" >> $O/graderr.log; IRRL_ENV_LIB=$PWD/$f timeout 600 python tools/lstm_grad_error.py 750 4096 48 2>/dev/null >> $O/graderr.log
      done ;;
    abmlp)
      # same-box A/B of the MlpPolicy update with / without the packed sample records, interleaved, three rounds
      rm -f $O/abmlp.log
      for r in 1 2 3; do for rec in 1 0; do
        IRRL_MLP_RECORDS=$rec timeout 300 python tools/ppo_bench.py --policy mlp --envs 4096 --iters 5 --cfg bp5_imitation.yaml 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('IRRL_MLP_RECORDS=$rec rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')" >> $O/abmlp.log
      done; done ;;
    abmlpw)
      # same-box A/B of the MlpPolicy gradient kernels: one wave per SIMD (IRRL_MLP_WAVES=4) against producer / consumer wave pairs (default)
      rm -f $O/abmlpw.log
      for r in 1 2; do for w in 4 8; do
        echo "IRRL_MLP_WAVES=$w" >> $O/abmlpw.log
        IRRL_MLP_WAVES=$w timeout 300 python tools/mlp_kernel_time.py 2>&1 | grep bf16x3 >> $O/abmlpw.log
      done; done
      for r in 1 2 3; do for w in 4 8; do
        IRRL_MLP_WAVES=$w timeout 300 python tools/ppo_bench.py --policy mlp --envs 4096 --iters 5 --cfg bp5_imitation.yaml 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('IRRL_MLP_WAVES=$w rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms', round(d['ppo_iters_per_sec'],3), 'it/s')" >> $O/abmlpw.log
      done; done ;;
    spread)
      V=high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants; rm -f $O/wave_spread.log
      IRRL_ENV_LIB=$PWD/$V/libirrl_env_prof.so timeout 300 python tools/wave_spread.py 2>/dev/null | tail -1 >> $O/wave_spread.log
      IRRL_ENV_LIB=$PWD/$V/libirrl_env_prof.so timeout 300 python tools/wave_spread.py --cfg default_cfg.yaml --sigma 1.0 2>/dev/null | tail -1 >> $O/wave_spread.log ;;
    ab) "$@" > $O/ab.log 2>&1; echo "rc=$?" >> $O/ab.log; break ;;
    *) echo "unknown stage $stage" >&2 ;;
  esac
done
echo done
