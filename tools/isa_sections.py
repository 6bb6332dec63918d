"""Instruction-count breakdown of one physics substep of the step kernel (16-lane layout) by source section.
Builds csrc/env_kernels.hip to assembly with -DIRRL_MARKS (section markers + scheduling barriers at the section borders, so
the total is a few percent above the production build) and counts VALU / SALU / other instructions between markers.
usage: python tools/isa_sections.py [--kernel NAME] [extra -D flags]   (default kernel: irrl_step_kernel_l16 = the published contact rule, shipped solver settings compiled in, run-time terrain test -- irrl_step_kernel_flat_l16: flat ground compiled in;
irrl_step_kernel_dir_l16 = the build's first rule, irrl_step_kernel_md_l16 = the published rule with the solver settings read at run time)"""
import os, re, subprocess, sys, tempfile, collections

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "high_speed_quadrupedal_locomotion_by_irrl_amd", "csrc", "env_kernels.hip")
out = os.path.join(tempfile.mkdtemp(), "marks.s")
kernel = "irrl_step_kernel_l16"
if "--kernel" in sys.argv:
    i = sys.argv.index("--kernel")
    kernel = sys.argv[i + 1]
    del sys.argv[i:i + 2]
sys.path.insert(0, root)
from high_speed_quadrupedal_locomotion_by_irrl_amd import build as B   # the product build's own flag lists
cmd = [B.hipcc()] + B.COMMON_FLAGS + ["-DIRRL_LANES_PER_ROBOT=16", "-DIRRL_MARKS"] + B.ENV_FLAGS + ["-S", "--cuda-device-only", "-o", out, src] + sys.argv[1:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
sec, counts, order = None, collections.OrderedDict(), []
inside = False
for line in open(out):
    t = line.strip()
    if t.split(";")[0].strip() == kernel + ":":
        inside = True
    elif inside and t.startswith(".Lfunc_end"):
        inside = False
    if not inside:
        continue
    m = re.match(r"; IRRL_MARK (\w+)", t)
    if m:
        sec = m.group(1)
        counts.setdefault(sec, collections.Counter())
        continue
    if sec is None or sec == "end" or not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "mem" if op.startswith(("ds_", "global_", "buffer_", "scratch_", "flat_")) else "other"
    counts[sec][kind] += 1
    if op.endswith("_dpp") or "_dpp" in t:
        counts[sec]["dpp"] += 1
    if op in ("v_readlane_b32", "v_writelane_b32"):
        counts[sec]["lane"] += 1
    if op.startswith("v_accvgpr"):
        counts[sec]["acc"] += 1
    if op.startswith("v_pk_"):
        counts[sec]["pk"] += 1
    if op in ("v_mov_b32_e32", "v_mov_b32_dpp", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_cndmask_b32_e32", "v_cndmask_b32_e64"):
        counts[sec]["mov_sel"] += 1
tot = collections.Counter()
print("%-17s %6s %6s %5s %5s %5s %7s %9s %8s" % ("section", "valu", "salu", "mem", "dpp", "pk", "mov/sel", "sgpr<->v", "agpr<->v"))
for k, c in counts.items():
    print("%-17s %6d %6d %5d %5d %5d %7d %9d %8d" % (k, c["valu"], c["salu"], c["mem"], c["dpp"], c["pk"], c["mov_sel"], c["lane"], c["acc"]))
    tot.update(c)
print("%-17s %6d %6d %5d %5d %5d %7d %9d %8d" % ("total", tot["valu"], tot["salu"], tot["mem"], tot["dpp"], tot["pk"], tot["mov_sel"], tot["lane"], tot["acc"]))
print("(gs = the sweep loop, static -- the default pool's kernels carry the simultaneous-sweep loop only, irrl_step_kernel_md_l16 both solvers' loops;\n sgpr<->v: v_readlane / v_writelane of spilled SGPRs, agpr<->v: v_accvgpr moves -- both are VALU issue slots)")
