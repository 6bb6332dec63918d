#!/usr/bin/env python3
"""What was running beside the launches of one kernel?  From a rocprofv3 --kernel-trace CSV: for every launch of kernels whose name contains
<pattern> its duration (End - Start as the trace records it), how much of that span other kernels were running too, and which.  A small kernel
enqueued on a second stream beside a long one 'lasts' as long as it has to wait for free compute units -- its trace duration says when its
LAST workgroup finished, not how much work it was.
    python tools/kernel_overlap.py <..._kernel_trace.csv> <pattern>"""
import csv
import sys
from collections import Counter

path, pat = sys.argv[1], sys.argv[2]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
mine = [r for r in rows if pat in r[2]]
if not mine:
    raise SystemExit("no kernel matching %r" % pat)
alone, shared, beside = [], [], Counter()
for s, e, n in mine:
    ov = [(max(s, s2), min(e, e2), n2) for s2, e2, n2 in rows if e2 > s and s2 < e and not (s2 == s and e2 == e and n2 == n)]
    # union of the overlapping intervals
    iv = sorted((a, b) for a, b, _ in ov)
    covered, cur = 0, s
    for a, b in iv:
        a = max(a, cur)
        if b > a:
            covered += b - a
            cur = b
    d = e - s
    (shared if covered > 0.5 * d else alone).append(d / 1e3)
    for _, _, n2 in ov:
        beside[n2.split("(")[0][:60]] += 1
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
print("%d launches of *%s*: trace duration min %.1f / median %.1f / mean %.1f / max %.1f us" % (len(mine), pat, min((e - s) for s, e, _ in mine) / 1e3,
      med([(e - s) / 1e3 for s, e, _ in mine]), sum((e - s) for s, e, _ in mine) / 1e3 / len(mine), max((e - s) for s, e, _ in mine) / 1e3))
print("  %d launches ran (mostly) ALONE: median %.1f us, mean %.1f us" % (len(alone), med(alone), sum(alone) / max(1, len(alone))))
print("  %d launches spent more than half of their span BESIDE other kernels: median %.1f us, mean %.1f us" % (len(shared), med(shared), sum(shared) / max(1, len(shared))))
for n2, c in beside.most_common(6):
    print("    beside %-60s %d x" % (n2, c))
