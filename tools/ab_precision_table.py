#!/usr/bin/env python3
"""Table of the seeded A/B training runs of `tools/gpu.sh abprec` (config 3, 4096 envs; arms = arithmetic of the LSTM update's sequence kernels,
optionally the actor's and the critic's stacks apart; several seeds per arm).  Per window of 20 updates and per arm: mean over the seeds of the
episode reward and of the explained variance, each with the seeds' spread (max - min), and at the end the question the table exists for: is the
gap between two arms' means larger than the spread inside the arms?
Reads gpurun_out/abprec/train_<arm>_s<seed>.log (the learner's own log lines), writes the table to stdout.
    python tools/ab_precision_table.py [dir] > profiles/r06_ab_lstm_precision_seeds.log"""
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "abprec")
runs = {}       # arm -> seed -> rows
for fn in sorted(os.listdir(d)):
    m = re.match(r"train_(.+)_s(\d+)\.log$", fn)
    if not m:
        continue
    rows = []
    for line in open(os.path.join(d, fn)):
        f = dict(g.groups() for g in re.finditer(r"(\w+) ([-+0-9.eEnaN]+)(?: \||$)", line))
        if "nupdates" in f:
            rows.append((int(f["nupdates"]), float(f["ep_reward_mean"]), float(f["explained_variance"]), float(f.get("iters_per_sec", "nan")),
                         float(f.get("value_loss", "nan"))))
    if rows:
        runs.setdefault(m.group(1), {})[int(m.group(2))] = rows
if not runs:
    raise SystemExit("no train_<arm>_s<seed>.log under %s" % d)
arms = list(runs)
seeds = sorted(set.intersection(*[set(v) for v in runs.values()]))
n = min(len(runs[a][s]) for a in arms for s in seeds)
win = 20
print("# config 3 (default_cfg.yaml: LSTM policy, full reward, ObsNoise 2, StochasticDynamics), 4096 envs x 750 steps per update, lr 1e-3, %d updates;" % n)
print("# arms %s; seeds %s (yaml `seed` and environment.seedd both set to it).  Cells: mean over the seeds [max - min over the seeds], windows of %d updates."
      % (", ".join(arms), " ".join(map(str, seeds)), win))
print("%-10s" % "updates" + "".join("%26s%22s" % (a + " reward", "expl.var") for a in arms))
table = {}
for lo in range(0, n, win):
    hi = min(n, lo + win)
    cells = []
    for a in arms:
        rs, evs = [], []
        for s in seeds:
            seg = runs[a][s][lo:hi]
            rs.append(sum(x[1] for x in seg) / len(seg))
            evs.append(sum(x[2] for x in seg) / len(seg))
        table[(lo, a)] = (rs, evs)
        cells.append("%17.1f [%6.1f]%13.3f [%.3f]" % (sum(rs) / len(rs), max(rs) - min(rs), sum(evs) / len(evs), max(evs) - min(evs)))
    print("%-10s" % ("%d-%d" % (lo + 1, hi)) + "".join(cells))
last = max(lo for lo, _ in table)
print("# per seed, final window (updates %d-%d):" % (last + 1, n))
for a in arms:
    rs, evs = table[(last, a)]
    print("#   %-22s reward %s   expl.var %s" % (a, " ".join("%7.1f" % r for r in rs), " ".join("%.3f" % e for e in evs)))
print("# explained-variance trend after update 140 (last window's mean minus the 121-140 window's), per seed:")
for a in arms:
    if (120, a) in table:
        d_ev = [e1 - e0 for e0, e1 in zip(table[(120, a)][1], table[(last, a)][1])]
        print("#   %-22s %s   mean %+.3f" % (a, " ".join("%+.3f" % v for v in d_ev), sum(d_ev) / len(d_ev)))
its = {a: sum(x[3] for s in seeds for x in runs[a][s][5:n]) / max(1, sum(len(runs[a][s][5:n]) for s in seeds)) for a in arms}
print("# learner's own clock, iterations per second (logging included): " + ", ".join("%s %.2f" % (a, its[a]) for a in arms))
