"""Summarise the rocprofv3 --pmc passes of `tools/gpu.sh pmc` (gpurun_out/pmc_env_*/<host>/<pid>_counter_collection.csv, newest
file per counter group) into profiles/<name>.json and profiles/pmc_summary_latest.json (read by bench.py for `traffic`).
FETCH_SIZE / WRITE_SIZE are in KB; the fetch correction is calibrated on the dword-per-lane copy kernel of the same pass
(MI355X_MICROARCH.md, HBM section).  The summary carries the irrl_version() of the library the counters were measured on
(gpurun_out/pmc_env_version.txt, written by tools/pmc_workload.py on the GPU box): bench.py reports `traffic` / `valu_issue` only
when that equals the library it is running.   usage: python tools/pmc_summarize.py r01_pmc_summary_step5"""
import csv, glob, json, os, statistics, sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
name = sys.argv[1] if len(sys.argv) > 1 else "pmc_summary"
per = {}
calib = {}
pers = {}
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", "pmc_env_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    f = max(files, key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        k, c, v = r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
        if k.startswith("irrl_step_kernel"):
            per.setdefault(k, {}).setdefault(c, []).append(v)
        elif k.startswith("irrl_steps_persistent_kernel"):
            pers.setdefault(c, []).append(v)
        elif k.startswith("irrl_calib_copy_kernel"):
            calib.setdefault(c, []).append(v)
step_name = max(per, key=lambda k: len(next(iter(per[k].values()))))
cnt = {c: {"median_per_launch": statistics.median(v), "launches": len(v)} for c, v in sorted(per[step_name].items())}
copy_bytes = 268435456
cal = {"kernel": "irrl_calib_copy_kernel (one dword per lane)", "bytes_read": copy_bytes, "bytes_written": copy_bytes,
       "FETCH_SIZE_KB": statistics.median(calib["FETCH_SIZE"]), "WRITE_SIZE_KB": statistics.median(calib["WRITE_SIZE"])}
cal["fetch_correction"] = copy_bytes / (cal["FETCH_SIZE_KB"] * 1024.0)
cal["write_correction"] = copy_bytes / (cal["WRITE_SIZE_KB"] * 1024.0)
rd = cnt["FETCH_SIZE"]["median_per_launch"] * 1024.0 * cal["fetch_correction"]
wr = cnt["WRITE_SIZE"]["median_per_launch"] * 1024.0 * cal["write_correction"]
m = lambda c: cnt[c]["median_per_launch"]
ver_file = os.path.join(root, "gpurun_out", "pmc_env_version.txt")
out = {"envs": 4096, "kernel": step_name, "library": open(ver_file).read().strip() if os.path.exists(ver_file) else None,
       "source": "rocprofv3 --pmc <group> --kernel-trace, one pass per group (tools/gpu.sh pmc), 4096 envs, 400 launches (100 landing pre-roll + 300), bp5_imitation.yaml, Philox action stream; per-launch medians",
       "calibration": cal, "counters": cnt,
       "hbm_bytes_per_launch": {"read": rd, "written": wr, "total": rd + wr, "algorithmic": 1521 * 4096},
       "derived": {"valu_insts_per_wave": m("SQ_INSTS_VALU") / m("SQ_WAVES"), "salu_insts_per_wave": m("SQ_INSTS_SALU") / m("SQ_WAVES"),
                   "cycles_per_valu_inst": 4.0 * m("SQ_WAVE_CYCLES") / m("SQ_INSTS_VALU"),
                   "valu_active_fraction": m("SQ_ACTIVE_INST_VALU") / m("SQ_WAVE_CYCLES"),
                   "wait_inst_any_fraction": m("SQ_WAIT_INST_ANY") / m("SQ_WAVE_CYCLES"),
                   "l2_hit_rate": m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum")),
                   "note": "SQ_WAVE_CYCLES etc. are in units of 4 clocks (quad-cycles)"}}
if pers:      # the multi-step persistent kernel: tools/pmc_workload.py issues launches of 100 steps each
    PERSIST_STEPS = 100
    pm = {c: statistics.median(v) for c, v in pers.items()}
    prd, pwr = pm["FETCH_SIZE"] * 1024.0 * cal["fetch_correction"], pm["WRITE_SIZE"] * 1024.0 * cal["write_correction"]
    out["persistent"] = {"kernel": "irrl_steps_persistent_kernel_flat_l16", "steps_per_launch": PERSIST_STEPS, "launches": len(pers["FETCH_SIZE"]),
                         "outputs_kept": True,   # tools/pmc_workload.py: every step's ob / reward / done / extraInfo stored to its own row (173 B per env-step)
                         "hbm_bytes_per_launch": {"read": prd, "written": pwr, "total": prd + pwr},
                         "hbm_bytes_per_step": {"read": prd / PERSIST_STEPS, "written": pwr / PERSIST_STEPS, "total": (prd + pwr) / PERSIST_STEPS, "algorithmic": 1521 * 4096},
                         "derived": {"valu_insts_per_wave_per_step": pm["SQ_INSTS_VALU"] / pm["SQ_WAVES"] / PERSIST_STEPS,
                                     "cycles_per_valu_inst": 4.0 * pm["SQ_WAVE_CYCLES"] / pm["SQ_INSTS_VALU"],
                                     "valu_active_fraction": pm["SQ_ACTIVE_INST_VALU"] / pm["SQ_WAVE_CYCLES"],
                                     "wait_inst_any_fraction": pm["SQ_WAIT_INST_ANY"] / pm["SQ_WAVE_CYCLES"],
                                     "l2_hit_rate": pm["TCC_HIT_sum"] / (pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"])}}
for fn in (name + ".json", "pmc_summary_latest.json"):
    json.dump(out, open(os.path.join(root, "profiles", fn), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "hbm_bytes_per_launch", "derived", "persistent") if k in out}, indent=1))
