"""Copy what a `tools/gpu.sh box smoke tests bench bench3 solvers prof pmc ppo sweep profppo` call merged into gpurun_out/ into
profiles/ under this round's names, and summarise the PMC passes.   usage: python tools/collect_evidence.py r03"""
import glob, json, os, shutil, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
G, P = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
def line(f):
    for l in open(f):
        if l.startswith("{"):
            return l
for src, dst in (("bench.log", "bench.json"), ("bench_driver_style.log", "bench_driver_style.json"), ("bench_default_cfg.log", "bench_train.json"),
                 ("bench_bp5_terrain.log", "bench_terrain.json")):
    if os.path.exists(os.path.join(G, src)):
        open(os.path.join(P, "%s_%s" % (tag, dst)), "w").write(line(os.path.join(G, src)))
for src, dst in (("pytest_gpu.log", "pytest_gpu.log"), ("smoke.log", "smoke.log"), ("box.log", "box.log"), ("sweep.log", "sweep_envs_per_gpu.log"),
                 ("ppo_lstm.log", "ppo_lstm_4096x750.log"), ("ppo_mlp.log", "ppo_mlp_4096x750.log"), ("solvers.log", "bench_by_contact_solver.log")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, "%s_%s" % (tag, dst)))
for pat, dst in (("prof_bench/*/*kernel_stats.csv", "bench_kernel_stats.csv"), ("prof_ppo/*/*kernel_stats.csv", "ppo_lstm_kernel_stats_2iters_2epochs.csv"),
                 ("prof_mlp/*/*kernel_stats.csv", "ppo_mlp_kernel_stats.csv")):
    st = sorted(glob.glob(os.path.join(G, pat)), key=os.path.getmtime)
    if st:
        shutil.copy(st[-1], os.path.join(P, "%s_%s" % (tag, dst)))
if glob.glob(os.path.join(G, "pmc_env_*")):
    print(subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summarize.py"), tag + "_pmc_summary"], capture_output=True, text=True).stdout[-600:])
for f in sorted(glob.glob(os.path.join(P, tag + "_bench*.json"))):
    d = json.loads(open(f).read())
    print(os.path.basename(f), round(d["value"] / 1e6, 2), "M env-steps/s", round(d["roofline"].get("avg_step_us", d["roofline"]["avg_launch_us"]), 2), "us per step (", d["roofline"].get("steps_per_launch", 1), "per launch)  fp32", round((d.get("roofline_fp32") or d["roofline"])["frac"], 4),
          "traffic", d["roofline"]["traffic"], "| per-call", d.get("per_step_call") and round(d["per_step_call"]["value"] / 1e6, 1), "| ppo", d.get("ppo", {}).get("ppo_iters_per_sec"),
          "| cpu", d.get("cpu_baseline", {}).get("value"))
