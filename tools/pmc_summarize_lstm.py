"""Summarise the rocprofv3 --pmc passes of `tools/gpu.sh pmclstm` (one PPO iteration with ONE epoch of the LSTM policy at
4096 x 750; gpurun_out/pmc_lstm_*/<host>/<pid>_counter_collection.csv) into profiles/<name>.json: per-kernel medians per launch
and the derived fractions DESIGN.md section 7 quotes.  usage: python tools/pmc_summarize_lstm.py r03_pmc_lstm_kernels
With a second argument `mlp`: the passes of `tools/gpu.sh pmcmlp` (gpurun_out/pmc_mlp_*: one PPO iteration, one epoch, of the MlpPolicy learner),
kernels irrl_mlp_ppo_kernel<0|1> / mlp_policy_step_kernel:  python tools/pmc_summarize_lstm.py r03_pmc_mlp_kernels mlp"""
import csv, glob, json, os, statistics, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
name = sys.argv[1] if len(sys.argv) > 1 else "pmc_lstm_kernels"
which = sys.argv[2] if len(sys.argv) > 2 else "lstm"
tokens = ("lstm_seq", "ppo_heads", "policy_step", "rollout_persistent") if which == "lstm" else ("irrl_mlp_ppo", "irrl_mlp_pack", "mlp_policy_step", "irrl_sum_rows", "rollout_persistent")
per = {}
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_*" % which))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = r["Kernel_Name"]
        if any(t in k for t in tokens):
            per.setdefault(k.replace("void ", "").split("(")[0], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
src = ("(rollout of 750 policy steps + one epoch: 4 forward, 4 backward sequence kernels, heads + loss); medians per launch; "
       "counters are collected with kernels serialised, so co-residency of the two stacks' kernels is NOT in these numbers") if which == "lstm" else (
       "(rollout of 750 policy steps + one epoch = 4 minibatches of 768 k samples: 4 launches of each gradient kernel); medians per launch")
out = {"source": "rocprofv3 --pmc <group> --kernel-trace, one pass per group (tools/gpu.sh pmc%s): tools/ppo_bench.py --policy %s --envs 4096 --iters 1 --epochs 1 " % (which, which) + src,
       "units": "SQ_*_CYCLES in quad-cycles (4 clocks) summed over waves / SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs; FETCH/WRITE_SIZE in KB "
                "(FETCH_SIZE x 2 on gfx950 for wide loads, MI355X_MICROARCH.md)", "kernels": {}}
for k, cs in sorted(per.items()):
    m = {c: statistics.median(v) for c, v in cs.items()}
    n = {c: len(v) for c, v in cs.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                      # kernel duration in clocks (per-XCD counter summed over 8 XCDs)
    d = {"launches_seen": max(n.values()), "counters": {c: m[c] for c in sorted(m)}}
    if cyc > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        d["derived"] = {
            "kernel_clocks": cyc,
            "mfma_busy_fraction_of_all_1024_simds": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
            "f32_mfma_flops": m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0,
            "valu_insts_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"], "mfma_insts_per_wave": m["SQ_INSTS_MFMA"] / m["SQ_WAVES"],
            "lds_insts_per_wave": m["SQ_INSTS_LDS"] / m["SQ_WAVES"], "salu_insts_per_wave": m["SQ_INSTS_SALU"] / m["SQ_WAVES"],
            "valu_active_fraction_of_wave_life": m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"],
            "lds_active_fraction_of_wave_life": m["SQ_ACTIVE_INST_LDS"] / m["SQ_WAVE_CYCLES"],
            "wait_inst_any_fraction_of_wave_life": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
            "wait_inst_lds_fraction_of_wave_life": m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"],
            "lds_bank_conflict_cycles_per_lds_active_cycle": m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_ACTIVE_INST_LDS"], 1.0),
            "hbm_bytes": {"read": m.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0, "written": m.get("WRITE_SIZE", 0.0) * 1024.0}}
    out["kernels"][k] = d
json.dump(out, open(os.path.join(root, "profiles", name + ".json"), "w"), indent=1)
for k, d in out["kernels"].items():
    if "derived" in d:
        x = d["derived"]
        print("%-48s clocks %.3g  MFMA-busy %.3f  VALU/wave %.0f  MFMA/wave %.0f  valu-active %.2f  wait-any %.2f  wait-lds %.3f  bank-conflict/lds %.2f" % (
            k[:48], x["kernel_clocks"], x["mfma_busy_fraction_of_all_1024_simds"], x["valu_insts_per_wave"], x["mfma_insts_per_wave"],
            x["valu_active_fraction_of_wave_life"], x["wait_inst_any_fraction_of_wave_life"], x["wait_inst_lds_fraction_of_wave_life"], x["lds_bank_conflict_cycles_per_lds_active_cycle"]))
