#!/usr/bin/env python3
"""Rewrite the two generated tables of DESIGN.md section 6 ("Where the kernel stands", "PPO iterations") and the figures of the default bench line
quoted in section 3.1 from profiles/rNN_bench*.json, so that the document quotes exactly what is committed:  python tools/refresh_design_tables.py r04"""
import json, os, re, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = lambda n: json.load(open(os.path.join(root, "profiles", "%s_%s.json" % (tag, n))))
b, dr, g8 = P("bench"), P("bench_driver_style"), P("bench_gloo8_one_device")
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
chk, pr = b["steady_state_check"], dr["config"]["launch_probe_us_per_step"]
a, e = s.index("### Where the kernel stands (round 4; history in Appendix A)"), s.index("PPO iterations (one iteration = 750-step rollout of 4096 envs")
med = chk["us_per_step_min_median_max"]
s = s[:a] + '''### Where the kernel stands (round 4; history in Appendix A)
| | µs per `env.step()` | env-steps/s | evidence |
|---|---|---|---|
| config 2 (`bp5_imitation.yaml`), 3000 steps as ONE persistent launch (`bench.py` default line) | **%.2f** | **%.1f M** | `profiles/%s_bench.json` (`steady_state_check.persistent_us_per_step` %.2f over 2000 steps) |
| the same, one launch per step, windows of 2000-3000 steps | **%.2f** | **%.1f M** | `profiles/%s_bench.json` `steady_state_check`: %.1f / %.1f / %.1f min / median / max over five windows |
| the driver's bracket (`--steps 20 --warmup 5`): one persistent launch of 20 steps | %.1f kernel, %.1f wall | %.1f M (106-113 M over the round's runs) | `profiles/%s_bench_driver_style.json` (`launch_probe_us_per_step`: persistent %.1f, rows %.1f, graph %.1f, python %.1f) |
| training config / terrain (config 5 ingredients), persistent launch | 34.6 / 36.5 | 118.3 M / 112.2 M | `%s_bench_train.json`, `%s_bench_terrain.json` |
| one `FlexibleGymEnv.step()` call per step: ctypes class / compiled pybind11 class | %.1f / %.1f | %.1f M / %.1f M | `per_step_call`, `per_step_call_compiled` |
| numpy (`_host`) boundary, PCIe inclusive | 147.7 | 27.7 M | `%s_sweep_envs_per_gpu.log` |
| 8 ranks × 4096 envs on ONE device over gloo (configs 4 / 5 at their real size without the node) | %.0f per rank-step | %.1f M (terrain 109.7 M) | `%s_bench_gloo8*_one_device.json` |
| CPU baseline (kind "port": the f64 oracle, OpenMP dynamic over envs like `VEC:273`) | | 0.58-0.66 M on the box's 16 cgroup cores (EPYC 9575F), 43-44 k on one thread | `cpu_baseline` |

Roofline of the default line: FP32 VALU view (the binding one, SURVEY §8d) 1.18×10⁵ flop × 4096 ÷ %.2f µs = %.1f TFLOP/s = **%.3f of 157.3**; HBM view
1521 B × 4096 ÷ %.2f µs = %.0f GB/s = %.3f of 8 TB/s; counter traffic %.2f MB per step (the pool lives in the L2s across a wave's own steps; one launch
per step: 8.80 MB per launch = 1.41 × algorithmic).  At 131 072 envs (4 lanes per robot) 393 M env-steps/s = 0.295 of the FP32 peak.  The flop constant is the
instrumented oracle's count for the reference order of operations (Gauss-Seidel + published rule, 1.6 sweeps per substep), kept fixed across
builds as SURVEY §8d asks.

''' % (b["roofline"]["avg_step_us"], b["value"] / 1e6, tag, chk["persistent_us_per_step"], med[1], 4096 / med[1], tag, med[0], med[1], med[2],
       dr["roofline"]["avg_step_us"], dr["ms_per_step"] * 1e3, dr["value"] / 1e6, tag, pr["persistent"], pr["rows"], pr["graph"], pr["python"], tag, tag,
       b["per_step_call"]["us_per_step"], b["per_step_call_compiled"]["us_per_step"], b["per_step_call"]["value"] / 1e6, b["per_step_call_compiled"]["value"] / 1e6, tag,
       g8["ms_per_step"] * 1e3, g8["value"] / 1e6, tag,
       b["roofline"]["avg_step_us"], b["roofline_fp32"]["achieved"], b["roofline_fp32"]["frac"], b["roofline"]["avg_step_us"], b["roofline"]["achieved"], b["roofline"]["frac"],
       (b["roofline"]["traffic"] or 0.0) / b["roofline"]["steps_per_launch"] / 1e6) + s[e:]
a, e = s.index("| CustomLSTMPolicy 2×48 + 2×48 (config 3), update on the bf16 matrix cores"), s.index("## 7. Learner (rows 20-23) and multi-GPU (row e)")
pl, pm, dl, dm = b["ppo"], b["ppo_mlp"], dr["ppo"], dr["ppo_mlp"]
lo, hi = (lambda k, x, y: 1e3 * min(x[k], y[k])), (lambda k, x, y: 1e3 * max(x[k], y[k]))
s = s[:a] + '''| CustomLSTMPolicy 2×48 + 2×48 (config 3), update on the bf16 matrix cores (`bf16x3`, §3.3), persistent rollout (§3.2) | %.1f-%.1f ms | **%.1f-%.1f ms** (the two committed lines; 101.5-113.5 over the round's boxes and builds) | **%.2f-%.2f** (6.28-6.85 over the round's boxes and builds) | 47 + 157 ms, 4.9 |
| same, `IRRL_LSTM_PRECISION=bf16x6` / `f32` | 45.5 ms | 146.8 / 158.3 ms | 5.20 / 4.91 | |
| MlpPolicy [64, 64] (config 2): persistent rollout (§3.2) + bf16x3 gradient kernels (§3.5) | **%.1f-%.1f ms** | **%.1f-%.1f ms** | **%.1f-%.1f** | 39 + 32 ms, 14.0 |
| same, `IRRL_MLP_ROLLOUT=direct` / `IRRL_MLP_PRECISION=f32` (same box A/B) | 40.8 ms | 27.2 ms | 16.7 / 14.7 (20.1 with both new) | |
| 8 ranks on one device over gloo, LSTM / MLP (collectives in the loop; eight processes time-slice one GPU and every all-reduce goes through the host) | | | %.2f / %.2f | |

''' % (lo("rollout_s", pl, dl), hi("rollout_s", pl, dl), lo("update_s", pl, dl), hi("update_s", pl, dl), min(pl["ppo_iters_per_sec"], dl["ppo_iters_per_sec"]), max(pl["ppo_iters_per_sec"], dl["ppo_iters_per_sec"]),
       lo("rollout_s", pm, dm), hi("rollout_s", pm, dm), lo("update_s", pm, dm), hi("update_s", pm, dm), min(pm["ppo_iters_per_sec"], dm["ppo_iters_per_sec"]), max(pm["ppo_iters_per_sec"], dm["ppo_iters_per_sec"]),
       g8["ppo"]["ppo_iters_per_sec"], g8["ppo_mlp"]["ppo_iters_per_sec"]) + s[e:]
# section 3.1's quotes of the default line
s = re.sub(r"\*\*[0-9.]+ µs per step = [0-9.]+ M env-steps/s\*\* at K = 3000 \([0-9.]+ µs at K = 2000;", "**%.2f µs per step = %.1f M env-steps/s** at K = 3000 (%.1f µs at K = 2000;" % (b["roofline"]["avg_step_us"], b["value"] / 1e6, chk["persistent_us_per_step"]), s)
s = re.sub(r"1\.18×10⁵ flop × 4096 ÷ [0-9.]+ µs = [0-9.]+ TFLOP/s = \*\*[0-9.]+ of 157\.3\*\* \(persistent launch", "1.18×10⁵ flop × 4096 ÷ %.2f µs = %.1f TFLOP/s = **%.3f of 157.3** (persistent launch" % (b["roofline"]["avg_step_us"], b["roofline_fp32"]["achieved"], b["roofline_fp32"]["frac"]), s)
s = re.sub(r"`roofline` object carries\): 1521 B × 4096 ÷ [0-9.]+ µs = [0-9.]+ GB/s = [0-9.]+ of 8 TB/s", "`roofline` object carries): 1521 B × 4096 ÷ %.2f µs = %.0f GB/s = %.3f of 8 TB/s" % (b["roofline"]["avg_step_us"], b["roofline"]["achieved"], b["roofline"]["frac"]), s)
s = re.sub(r"and a step costs the wave its own time\.  [0-9.]+ µs per step at K = 3000", "and a step costs the wave its own time.  %.2f µs per step at K = 3000" % b["roofline"]["avg_step_us"], s)
open(p, "w").write(s)
print("default line %.2f us / %.1f M; driver-style %.1f M; ppo %.2f / %.2f" % (b["roofline"]["avg_step_us"], b["value"] / 1e6, dr["value"] / 1e6, pl["ppo_iters_per_sec"], pm["ppo_iters_per_sec"]))
