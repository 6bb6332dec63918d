#!/usr/bin/env python3
"""ISA post-pass for the env kernels (gfx950): fold `v_mov_b32_dpp vT, vS <ctrl>` into the `v_fmac_f32_e32 vD, vT, vY` that
consumes it -> `v_fmac_f32_dpp vD, vS, vY <ctrl>`.

Why a post-pass: LLVM's DPP combine runs before register allocation, where an FMA is still the three-address V_FMA_F32_e64
(no DPP encoding on gfx9); the two-address v_fmac_f32_e32 -- which HAS a DPP form -- only appears after allocation
(SIShrinkInstructions).  So the compiler folds DPP moves into v_add / v_mul / v_sub but never into an FMA, and every
`acc += lane_exchange(x) * y` of the kernels costs two issue slots.  Measured on MI355X (tools/microbench/valu_issue.hip): one wave
per SIMD issues v_mov_b32_dpp + v_fma_f32 in 5.4 ns but v_fmac_f32_dpp in 2.3 ns.

Safety rules (all checked on the final assembly, inside one basic block):
  * vT is written by the mov, read ONLY by that one fmac (as src0 or src1, not both, not as the accumulator) and then dead:
    it is overwritten before any other read in the same block (anything else: no fold);
  * neither vS nor vT is written between the mov and the fmac (when vS == vT the mov itself is the only writer);
  * EXEC is not written and no label / branch / s_barrier lies between the two (same lanes active at both places);
  * DPP read-after-write hazard (VALU write of vS -> DPP read needs 2 wait states, VALU write of EXEC -> 5): the pass looks at
    the instructions in front of the new v_fmac_f32_dpp and inserts `s_nop` only where needed.
usage: dpp_fold.py in.s out.s   (prints statistics)"""
import re
import sys

RE_MOV = re.compile(r"^\s*v_mov_b32_dpp\s+(v\d+),\s*(v\d+)\s+(.*)$")
RE_FMAC = re.compile(r"^\s*v_fmac_f32_e32\s+(v\d+),\s*(v\d+),\s*(v\d+)\s*$")
RE_VREG = re.compile(r"\bv(\d+)\b")
RE_VRANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
RE_ARANGE = re.compile(r"\ba\[(\d+):(\d+)\]")


def regs_of(tok):
    """vector registers named in an operand string"""
    out = set()
    for m in RE_VRANGE.finditer(tok):
        out.update("v%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1))
    for m in RE_VREG.finditer(tok):
        out.add("v" + m.group(1))
    return out


def split_inst(line):
    """-> (opcode, dst operand string, src operand string) of an instruction line, or None for directives / labels / comments"""
    t = line.split(";")[0].strip()
    if not t or t.startswith((".", "//")) or t.endswith(":"):
        return None
    parts = t.split(None, 1)
    op = parts[0]
    ops = parts[1] if len(parts) > 1 else ""
    if "," in ops:
        dst, src = ops.split(",", 1)
    else:
        dst, src = ops, ""
    return op, dst, src


NO_DST = ("v_cmp", "v_cmpx", "s_", "global_store", "buffer_store", "scratch_store", "ds_write", "flat_store", "v_nop", "s_nop", "v_readlane", "v_readfirstlane")


def writes_reads(line):
    """(vector registers written, vector registers read) -- conservative: unknown forms count every register as both"""
    si = split_inst(line)
    if si is None:
        return set(), set()
    op, dst, src = si
    if op.startswith(("global_load", "buffer_load", "scratch_load", "ds_read", "flat_load")):
        return regs_of(dst), regs_of(src)
    if op.startswith(NO_DST):
        allr = regs_of(dst) | regs_of(src)
        if op.startswith(("v_readlane", "v_readfirstlane")):
            return set(), regs_of(src)
        return set(), allr
    w = regs_of(dst)
    r = regs_of(src)
    if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_mfma")) or "_dpp" in op and False:
        r |= w   # accumulate in place
    if op.startswith(("v_writelane", "v_mov_b32_dpp", "v_add_f32_dpp", "v_mul_f32_dpp", "v_sub_f32_dpp", "v_subrev_f32_dpp", "v_add_u32_dpp")) and "bound_ctrl" not in line:
        r |= w   # DPP without bound_ctrl keeps the old value in disabled lanes
    if op.startswith("v_div_fmas") or op.startswith("v_cndmask"):
        pass
    return w, r


def is_block_end(line):
    t = line.split(";")[0].strip()
    if not t:
        return False
    if t.endswith(":") and not t.startswith("."):
        return True
    if re.match(r"^\.L\w+:", t):
        return True
    op = t.split()[0]
    return op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_setpc", "s_swappc", "s_call"))


def writes_exec(line):
    t = line.split(";")[0]
    si = split_inst(line)
    if si is None:
        return False
    op, dst, _ = si
    return "exec" in dst or op.startswith("v_cmpx") or "saveexec" in op


def is_valu(line):
    si = split_inst(line)
    return si is not None and si[0].startswith("v_")


def wait_states(line):
    si = split_inst(line)
    if si is None:
        return 0
    if si[0] == "s_nop":
        try:
            return int(si[1].strip(), 0) + 1
        except ValueError:
            return 1
    return 1


def fold(lines):
    n = len(lines)
    removed = set()
    replaced = {}
    nops = {}
    stats = dict(mov_dpp=0, folded=0, multi_use=0, not_fmac=0, live_out=0, clobbered=0, hazard_nops=0)
    for i, ln in enumerate(lines):
        m = RE_MOV.match(ln.split(";")[0])
        if not m:
            continue
        stats["mov_dpp"] += 1
        vT, vS, ctrl = m.group(1), m.group(2), m.group(3).strip()
        if "bound_ctrl" not in ctrl or "row_mask:0xf" not in ctrl or "bank_mask:0xf" not in ctrl:
            stats["not_fmac"] += 1
            continue
        # find the first reader of vT
        j = i + 1
        use = None
        ok = True
        while j < n:
            if is_block_end(lines[j]) or writes_exec(lines[j]):
                ok = False
                break
            w, r = writes_reads(lines[j])
            if vT in r:
                use = j
                break
            if vT in w or (vS in w):
                ok = False
                break
            j += 1
        if not ok or use is None:
            stats["clobbered"] += 1
            continue
        f = RE_FMAC.match(lines[use].split(";")[0])
        if not f:
            stats["not_fmac"] += 1
            continue
        vD, a, b = f.group(1), f.group(2), f.group(3)
        if vD == vT or (a == vT and b == vT) or (vS == vD and vS != vT):
            stats["not_fmac"] += 1
            continue
        other = b if a == vT else a
        if other == vT:
            stats["not_fmac"] += 1
            continue
        # vT must be dead after the fmac: overwritten before any read, inside this block
        k = use + 1
        dead = False
        while k < n:
            if is_block_end(lines[k]):
                break
            w, r = writes_reads(lines[k])
            if vT in r:
                break
            if vT in w:
                dead = True
                break
            k += 1
        if not dead:
            stats["live_out" if k >= n or is_block_end(lines[k]) else "multi_use"] += 1
            continue
        # when vS == vT the mov overwrote its own source: after removal the register still holds the un-permuted value, which
        # is what the DPP form reads -- but then nobody else may have expected the permuted value (checked: single reader)
        # hazard: VALU writes of vS within the 2 wait states in front of the fmac
        need = 0
        ws = 0
        k = use - 1
        while k > i and ws < 2:
            if k in removed:
                k -= 1
                continue
            si = split_inst(lines[k])
            if si is None:
                k -= 1
                continue
            w, _ = writes_reads(lines[k])
            if is_valu(lines[k]) and vS in w:
                need = max(need, 2 - ws)
            ws += wait_states(lines[k])
            k -= 1
        if k <= i and ws < 2:
            # the window reaches past the removed mov: look further up for writers of vS
            kk = i - 1
            while kk >= 0 and ws < 2:
                if kk in removed:
                    kk -= 1
                    continue
                if is_block_end(lines[kk]):
                    need = max(need, 2 - ws)   # unknown predecessor: be safe
                    break
                si = split_inst(lines[kk])
                if si is None:
                    kk -= 1
                    continue
                w, _ = writes_reads(lines[kk])
                if is_valu(lines[kk]) and vS in w:
                    need = max(need, 2 - ws)
                ws += wait_states(lines[kk])
                kk -= 1
        removed.add(i)
        replaced[use] = "\tv_fmac_f32_dpp %s, %s, %s %s\n" % (vD, vS, other, ctrl)
        if need:
            nops[use] = need
            stats["hazard_nops"] += 1
        stats["folded"] += 1
    out = []
    for i, ln in enumerate(lines):
        if i in removed:
            continue
        if i in nops:
            out.append("\ts_nop %d\n" % (nops[i] - 1))
        out.append(replaced.get(i, ln))
    return out, stats


if __name__ == "__main__":
    src = open(sys.argv[1]).readlines()
    out, stats = fold(src)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").writelines(out)
    print(stats)
