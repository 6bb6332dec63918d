#!/usr/bin/env python3
"""Timeline of one PPO update from a rocprofv3 --kernel-trace CSV: for each optimisation epoch (delimited by the heads / loss
kernel) the wall time, the time at least one kernel was running, the idle gaps and the kernel sequence with start offsets.
    python tools/ppo_timeline.py <..._kernel_trace.csv> [--epoch K]"""
import csv
import sys


def short(n):
    for key in ("lstm_seq_bwd_x_kernel", "lstm_seq_fwd_x_kernel", "irrl_ppo_heads_loss", "lstm_policy_step", "irrl_step_kernel", "irrl_gae",
                "multi_tensor_apply", "reduce_kernel", "elementwise", "fillBuffer", "copyBuffer", "CatArray", "irrl_partial", "index_"):
        if key in n:
            tmpl = n[n.find("<"):n.find(">") + 1] if "lstm_seq" in n and "<" in n else ""
            return key + tmpl
    return n[:40]


def main():
    path = sys.argv[1]
    want = int(sys.argv[sys.argv.index("--epoch") + 1]) if "--epoch" in sys.argv else -2
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "irrl_ppo_heads_loss" in r[2]]
    if len(marks) < 3:
        raise SystemExit("need at least 3 heads/loss launches in the trace")
    print("heads/loss launches:", len(marks))
    # an epoch = from the end of heads kernel k-1 ... no: from the first kernel after the previous epoch's LAST kernel.  Use heads-to-heads.
    spans = [(rows[marks[k]][0], rows[marks[k + 1]][0]) for k in range(len(marks) - 1)]
    durs = [(b - a) / 1e6 for a, b in spans]
    print("heads-to-heads spans (ms):", " ".join("%.2f" % d for d in durs))
    a, b = spans[want]
    sel = [r for r in rows if r[0] >= a and r[0] < b]
    busy, cur_end, gaps = 0, a, []
    for s, e, n, st, q in sel:
        if s > cur_end:
            gaps.append((cur_end - a, s - cur_end))
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    print("epoch span %.3f ms, busy (union) %.3f ms, idle %.3f ms in %d gaps" % ((b - a) / 1e6, busy / 1e6, (b - a - busy) / 1e6, len(gaps)))
    print("largest gaps (offset ms, length us):", ", ".join("%.2f:%.0f" % (o / 1e6, g / 1e3) for o, g in sorted(gaps, key=lambda x: -x[1])[:12]))
    print("%10s %10s  %-6s %s" % ("start ms", "dur us", "queue", "kernel"))
    for s, e, n, st, q in sel:
        if e - s > 20000 or "lstm" in n or "heads" in n:
            print("%10.3f %10.1f  %-6s %s" % ((s - a) / 1e6, (e - s) / 1e3, q, short(n)))
    small = [r for r in sel if r[1] - r[0] <= 20000 and "lstm" not in r[2] and "heads" not in r[2]]
    print("+ %d small kernels, %.3f ms in total" % (len(small), sum(r[1] - r[0] for r in small) / 1e6))


if __name__ == "__main__":
    main()
