#!/usr/bin/env python3
"""Overlay of the seeded A/B training runs of `tools/gpu.sh abprec` (config 3, 4096 envs, same seeds, the LSTM update's arithmetic at bf16x3 /
bf16x6 / f32): per update window the mean episode reward and explained variance of each run side by side, and the largest gap between the
curves.  Reads gpurun_out/abprec/train_<prec>.log (the learner's own log lines), writes the table to stdout.
    python tools/ab_precision_table.py [dir] > profiles/r05_ab_lstm_precision_training_curves.log"""
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "abprec")
runs = {}
for prec in ("bf16x3", "bf16x6", "f32"):
    path = os.path.join(d, "train_%s.log" % prec)
    if not os.path.exists(path):
        continue
    rows = []
    for line in open(path):
        f = dict(m.groups() for m in re.finditer(r"(\w+) ([-+0-9.eEnaN]+)(?: \||$)", line))
        if "nupdates" in f:
            rows.append((int(f["nupdates"]), float(f["ep_reward_mean"]), float(f["explained_variance"]), float(f.get("iters_per_sec", "nan"))))
    runs[prec] = rows
if not runs:
    raise SystemExit("no train_<prec>.log under %s" % d)
names = list(runs)
n = min(len(r) for r in runs.values())
print("# config 3 (default_cfg.yaml: LSTM policy, full reward, ObsNoise 2, StochasticDynamics), 4096 envs x 750 steps per update, seed 1, lr 1e-3,")
print("# %d updates from identical seeds; only lstm_fused.PRECISION differs.  Means over windows of 20 updates." % n)
print("%-12s" % "updates" + "".join("%14s%10s" % (p + " reward", "expl.var") for p in names) + "   it/s " + " / ".join(names))
win = 20
gap_r, gap_ev = 0.0, 0.0
for lo in range(0, n, win):
    hi = min(n, lo + win)
    cells, rs, evs, its = [], [], [], []
    for p in names:
        seg = runs[p][lo:hi]
        r = sum(x[1] for x in seg) / len(seg)
        ev = sum(x[2] for x in seg) / len(seg)
        it = sum(x[3] for x in seg) / len(seg)
        rs.append(r); evs.append(ev); its.append(it)
        cells.append("%14.1f%10.3f" % (r, ev))
    if lo >= 40:
        gap_r = max(gap_r, (max(rs) - min(rs)) / max(1.0, abs(sum(rs) / len(rs))))
        gap_ev = max(gap_ev, max(evs) - min(evs))
    print("%-12s" % ("%d-%d" % (lo + 1, hi)) + "".join(cells) + "   " + " / ".join("%.2f" % v for v in its))
print("# largest gap between the runs' window means after update 40: reward %.1f %% of the mean, explained variance %.3f" % (100 * gap_r, gap_ev))
print("# final window reward: " + ", ".join("%s %.1f" % (p, sum(x[1] for x in runs[p][n - win:n]) / win) for p in names))
