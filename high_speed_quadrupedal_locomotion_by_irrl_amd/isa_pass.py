"""ISA pass of the env-kernel build (gfx950): wait states for the hand-placed `v_fmac_f32_dpp` instructions.

Why it exists.  `acc += lane_exchange(x) * y` is the most common statement of the step kernel (CRBA columns, contact operators,
Delassus blocks, Gauss-Seidel coupling).  LLVM folds a DPP move into v_add / v_mul / v_sub but never into an FMA: its DPP
combine runs before register allocation, where an FMA is still the three-address V_FMA_F32_e64 (no DPP encoding on gfx9);
the two-address v_fmac_f32_e32, which HAS a DPP form, only appears after allocation.  So every such statement costs a
v_mov_b32_dpp AND a v_fmac -- 5.4 ns for one resident wave per SIMD instead of the 2.3 ns of a v_fmac_f32_dpp
(tools/microbench/valu_issue.hip, measured on MI355X).  The lane primitives (csrc/lanes_hip16.hpp: sub_bcast_fma, legs_rot_fma,
...) therefore emit v_fmac_f32_dpp as inline assembly.  The compiler's hazard recogniser cannot see through inline assembly,
and on gfx950 (measured: tools/microbench, "DPP RAW hazard") the hardware does NOT interlock, so this pass -- run by build.py on
the device assembly between the compiler and the assembler -- inserts exactly the `s_nop`s the ISA manual requires:

  * VALU writes a VGPR, a DPP instruction reads it as its DPP operand (src0): 2 wait states         (case: reader is ours)
  * our VALU instruction writes a VGPR, a compiler-generated DPP instruction reads it as src0: 2      (case: writer is ours)
  * VALU writes EXEC (v_cmpx), DPP instruction: 5 wait states
  * transcendental VALU (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) result read by the next VALU: 1 wait state
  * block boundaries: predecessors are not traced; an inline-assembly DPP instruction closer than 2 wait states to the start of
    its basic block, or an inline-assembly write closer than 2 wait states to the block's end, gets the missing wait states.
  * a VALU write of EXEC closer than 5 wait states to the END of its block gets the missing wait states there, so no successor
    block can start with a DPP instruction inside the window (predecessors are not traced: the writer pays, not every reader).
Compiler-generated pairs are the compiler's business and are left alone.  Every instruction counts one wait state, `s_nop N`
counts N + 1.  Inline assembly is recognised by the ;;#ASMSTART / ;;#ASMEND brackets the compiler prints around it; an
inline-assembly block must hold exactly ONE instruction (the wait states go in front of the bracket, so a hazard between two
instructions of one block could not be fixed): anything else is rejected.

FAIL CLOSED.  `verify` re-scans the OUTPUT of the pass with an independent, simpler rule set and raises `HazardError` if ANY DPP
instruction -- hand-placed or compiler-generated -- reads its DPP operand closer than 2 wait states behind a VALU write of that
register, closer than 5 behind a VALU write of EXEC, or (hand-placed ones) closer than 2 to the start of its block; or if a VALU
write of EXEC sits closer than 5 to a block end.  `process_file` (the build) always verifies: a missed hazard would be silently
wrong contact / CRBA numerics, so the build stops instead."""
import re
import sys

RE_VREG = re.compile(r"\bv(\d+)\b")
RE_VRANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_", "v_rcp_iflag", "v_exp_legacy", "v_log_legacy")


def _regs(tok):
    out = set()
    for m in RE_VRANGE.finditer(tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in RE_VREG.finditer(tok):
        out.add(int(m.group(1)))
    return out


class Inst(object):
    __slots__ = ("line", "op", "dst", "srcs", "ours", "text")

    def __init__(self, line, text, ours):
        self.line, self.text, self.ours = line, text, ours
        parts = text.split(None, 1)
        self.op = parts[0]
        ops = parts[1] if len(parts) > 1 else ""
        # strip DPP / modifier keywords after the operand list
        ops = re.split(r"\s+(?:quad_perm|row_|wave_|bound_ctrl|bank_mask|row_mask|op_sel|neg_|clamp|mul:|div:|offset|glc|slc|sc0|sc1|nt|off\b)", ops)[0]
        toks = [t.strip() for t in ops.split(",")] if ops.strip() else []
        self.dst = toks[0] if toks else ""
        self.srcs = toks[1:]

    @property
    def is_valu(self):
        return self.op.startswith("v_")

    @property
    def is_dpp(self):
        return "_dpp" in self.op

    @property
    def wait_states(self):
        if self.op == "s_nop":
            try:
                return int(self.dst, 0) + 1
            except ValueError:
                return 1
        return 1

    def writes(self):
        if not self.is_valu or self.op.startswith(("v_cmp", "v_nop", "v_readlane", "v_readfirstlane")):
            return set()
        if self.op.startswith("v_swap"):     # v_swap_b32 vA, vB writes both of its operands
            return _regs(self.dst) | (_regs(self.srcs[0]) if self.srcs else set())
        return _regs(self.dst)

    def reads(self):
        r = set()
        for s in self.srcs:
            r |= _regs(s)
        if self.op.startswith(("v_fmac", "v_mac", "v_pk_fmac")) or self.op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            r |= _regs(self.dst)
        return r

    def writes_exec_valu(self):
        return self.op.startswith("v_cmpx")


def _is_block_start(t):
    return bool(re.match(r"^[.\w$]+:", t)) and not t.startswith(";")


def _is_block_end(op):
    return op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_barrier"))


class HazardError(RuntimeError):
    pass


def _parse(lines):
    """-> per line: Inst, "BLOCK" (a label) or None.  Rejects inline-assembly blocks that hold more than one instruction."""
    insts = []
    in_asm, asm_count, asm_line = False, 0, 0
    for i, ln in enumerate(lines):
        t = ln.split("//")[0].strip()
        if t.startswith(";;#ASMSTART"):
            in_asm, asm_count, asm_line = True, 0, i
            insts.append(None)
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            insts.append(None)
            continue
        t = t.split(";")[0].strip()
        if not t or t.startswith(".") and not _is_block_start(t) or _is_block_start(t):
            insts.append("BLOCK" if _is_block_start(t) else None)
            continue
        if in_asm:
            asm_count += 1
            if asm_count > 1:
                raise HazardError("inline-assembly block at line %d holds more than one instruction: hazards inside it cannot be fixed" % (asm_line + 1))
        insts.append(Inst(i, t, in_asm))
    return insts


def verify(lines):
    """Independent re-scan (see the module docstring).  -> number of DPP instructions checked; raises HazardError."""
    insts = _parse(lines)
    problems, checked = [], 0
    for i, x in enumerate(insts):
        if not isinstance(x, Inst) or not x.is_valu:
            continue
        if x.writes_exec_valu():            # a VALU write of EXEC must be 5 wait states away from the end of its block
            ws, k = 0, i + 1
            while k < len(insts) and ws < 5:
                y = insts[k]
                if y == "BLOCK" or (isinstance(y, Inst) and _is_block_end(y.op)):
                    problems.append("line %d: VALU write of EXEC %d wait state(s) before a block end (line %d)" % (i + 1, ws, k + 1))
                    break
                if isinstance(y, Inst):
                    ws += y.wait_states
                k += 1
        if not (x.is_dpp and x.srcs):
            continue
        checked += 1
        dpp_src = _regs(x.srcs[0])
        ws, k, hit_start = 0, i - 1, False
        while k >= 0 and ws < 5:
            y = insts[k]
            if y == "BLOCK" or (isinstance(y, Inst) and _is_block_end(y.op)):
                hit_start = True
                break
            if isinstance(y, Inst):
                if y.is_valu and ws < 2 and (y.writes() & dpp_src):
                    problems.append("line %d: %s reads v%s as DPP operand %d wait state(s) behind line %d: %s" % (i + 1, x.op, sorted(y.writes() & dpp_src), ws, k + 1, y.text))
                if y.writes_exec_valu():
                    problems.append("line %d: %s %d wait state(s) behind a VALU write of EXEC (line %d)" % (i + 1, x.op, ws, k + 1))
                ws += y.wait_states
            k -= 1
        if x.ours and (hit_start or k < 0) and ws < 2:
            problems.append("line %d: hand-placed %s only %d wait state(s) behind the start of its block" % (i + 1, x.op, ws))
    if problems:
        raise HazardError("%d unresolved DPP / EXEC hazard(s) in the env-kernel assembly:\n  " % len(problems) + "\n  ".join(problems[:20]))
    return checked


def run(lines):
    """-> (new lines, statistics)"""
    insts = _parse(lines)          # per line: Inst, "BLOCK" or None
    n = len(lines)
    insert_before = {}   # line index -> wait states to add in front of it
    stats = dict(asm_dpp=0, nops_reader=0, nops_writer=0, nops_trans=0, nops_block=0, wait_states_added=0)

    def prev_window(i, need):
        """instructions in front of line i inside its basic block, nearest first, until `need` wait states are covered.
        -> (list of (inst, wait states between it and line i), reached_block_start, wait states seen)"""
        out, ws, k = [], 0, i - 1
        while k >= 0 and ws < need:
            x = insts[k]
            if x == "BLOCK":
                return out, True, ws
            if isinstance(x, Inst):
                if _is_block_end(x.op):
                    return out, True, ws
                out.append((x, ws))
                ws += x.wait_states + insert_before.get(k, 0)
            k -= 1
        return out, k < 0, ws

    for i in range(n):
        x = insts[i]
        if not isinstance(x, Inst) or not x.is_valu:
            continue
        need = 0
        why = None
        if x.is_dpp and x.srcs:
            dpp_src = _regs(x.srcs[0])
            if x.ours:
                stats["asm_dpp"] += 1
            win, hit_start, seen = prev_window(i, 5)
            for y, ws in win:
                if y.is_valu and (x.ours or y.ours) and ws < 2 and (y.writes() & dpp_src):
                    if 2 - ws > need:
                        need, why = 2 - ws, "nops_reader" if x.ours else "nops_writer"
                if y.writes_exec_valu() and x.ours and ws < 5 and 5 - ws > need:
                    need, why = 5 - ws, "nops_reader"
            if x.ours and hit_start and seen < 2 and 2 - seen > need:
                need, why = 2 - seen, "nops_block"
        if x.ours:
            # transcendental result consumed by our instruction right behind it
            win, _, _ = prev_window(i, 1)
            for y, ws in win:
                if ws == 0 and y.op.startswith(TRANS) and (y.writes() & x.reads()) and need < 1:
                    need, why = 1, "nops_trans"
        if need:
            # the nops go in front of the ;;#ASMSTART bracket when the instruction is ours
            at = i
            if x.ours:
                k = i - 1
                while k >= 0 and not lines[k].strip().startswith(";;#ASMSTART"):
                    k -= 1
                at = k if k >= 0 else i
            insert_before[at] = max(insert_before.get(at, 0), need)
            stats[why] += 1
    # our writes too close to a block end: pad the block end
    for i in range(n):
        x = insts[i]
        if not (isinstance(x, Inst) and x.ours and x.is_valu):
            continue
        ws, k = 0, i + 1
        while k < n and ws < 2:
            y = insts[k]
            if y == "BLOCK" or (isinstance(y, Inst) and _is_block_end(y.op)):
                at = k
                insert_before[at] = max(insert_before.get(at, 0), 2 - ws)
                stats["nops_block"] += 1
                break
            if isinstance(y, Inst):
                ws += y.wait_states
            k += 1
    # a VALU write of EXEC too close to a block end: the successors' DPP instructions cannot see it, so the writer pays
    for i in range(n):
        x = insts[i]
        if not (isinstance(x, Inst) and x.is_valu and x.writes_exec_valu()):
            continue
        ws, k = 0, i + 1
        while k < n and ws < 5:
            y = insts[k]
            if y == "BLOCK" or (isinstance(y, Inst) and _is_block_end(y.op)):
                insert_before[k] = max(insert_before.get(k, 0), 5 - ws)
                stats["nops_block"] += 1
                break
            if isinstance(y, Inst):
                ws += y.wait_states + insert_before.get(k, 0)
            k += 1
    out = []
    for i, ln in enumerate(lines):
        w = insert_before.get(i, 0)
        if w:
            out.append("\ts_nop %d\t\t\t\t; isa_pass: %d wait state(s) for a DPP / trans hazard the compiler cannot see\n" % (w - 1, w))
            stats["wait_states_added"] += w
        out.append(ln)
    return out, stats


def process_file(src, dst):
    with open(src) as f:
        lines = f.readlines()
    out, stats = run(lines)
    stats["dpp_verified"] = verify(out)      # fail closed: raises HazardError when anything is left
    with open(dst, "w") as f:
        f.writelines(out)
    return stats


if __name__ == "__main__":
    print(process_file(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "/dev/null"))
