#!/usr/bin/env python3
"""Instruction-class sequence of the loops of one kernel in a .s file (M mfma, v valu, T transcendental, a accvgpr move, r/w LDS
read/write, G/S global load/store, W waitcnt, B barrier, s scalar, n nop), run-length compressed.
    python tools/isa/loop_shape.py file.s kernel_symbol_prefix [n_loops]"""
import re, sys


def cls(l):
    for p, c in (("v_mfma", "M"), ("ds_read", "r"), ("ds_load", "r"), ("ds_write", "w"), ("ds_store", "w"), ("global_load", "G"), ("buffer_load", "G"),
                 ("global_store", "S"), ("s_waitcnt", "W"), ("s_barrier", "B"), ("v_accvgpr", "a"), ("v_exp", "T"), ("v_rcp", "T"), ("v_log", "T"),
                 ("v_sqrt", "T"), ("v_rsq", "T"), ("v_", "v"), ("s_nop", "n"), ("s_", "s")):
        if l.startswith(p):
            return c
    return "?"


def main():
    t = open(sys.argv[1]).read()
    name = sys.argv[2]
    nl = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    start = [m.start() for m in re.finditer(r"^" + re.escape(name) + r"\S*:", t, re.M)][0]
    b = t[start:t.index(".Lfunc_end", start)]
    lines = [l.strip() for l in b.splitlines() if l.strip() and not l.strip().startswith((";", "//"))]
    lab = {l[:-1]: k for k, l in enumerate(lines) if l.endswith(":") and l.startswith(".LBB")}
    loops = []
    for k, l in enumerate(lines):
        m = re.match(r"s_c?branch\w* (\.LBB\S+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < k:
            loops.append((k - lab[m.group(1)], lab[m.group(1)], k))
    loops.sort(reverse=True)
    for L, st, en in loops[:nl]:
        seq = "".join(cls(l) for l in lines[st:en + 1] if not l.endswith(":"))
        out, prev, cnt = [], None, 0
        for c in seq + "\0":
            if c == prev:
                cnt += 1
            else:
                if prev:
                    out.append(prev + (str(cnt) if cnt > 1 else ""))
                prev, cnt = c, 1
        print("loop lines %d..%d: %d instructions: %s" % (st, en, len(seq), " ".join("%s %d" % (c, seq.count(c)) for c in "MvTarwGSWBsn" if seq.count(c))))
        print(" ".join(out))


if __name__ == "__main__":
    main()
