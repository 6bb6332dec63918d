"""Basic-block structure (instruction counts between labels / branches) of one marked section of a step kernel: where the
contact sweep loop's instructions sit.  usage: python tools/isa_loop_blocks.py [--kernel NAME] [--section gs] [-D flags]"""
import os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "high_speed_quadrupedal_locomotion_by_irrl_amd", "csrc", "env_kernels.hip")
args = sys.argv[1:]
def opt(name, default):
    if name in args:
        i = args.index(name); v = args[i + 1]; del args[i:i + 2]; return v
    return default
kernel, section = opt("--kernel", "irrl_step_kernel_l16"), opt("--section", "gs")
lanes = "4" if kernel.endswith("_l4") else "16"
out = os.path.join(tempfile.mkdtemp(), "marks.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fno-signed-zeros", "-DIRRL_LANES_PER_ROBOT=" + lanes,
                "-DIRRL_MARKS", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-S", "--cuda-device-only", "-o", out, src] + args, check=True, stderr=subprocess.DEVNULL)
t = open(out).read()
i = t.index("\n" + kernel + ":")
k = t[i:t.index(".Lfunc_end", i)]
marks = [m.start() for m in re.finditer(r"; IRRL_MARK ", k)]
a = k.index("; IRRL_MARK " + section)
b = min(m for m in marks if m > a)
cnt = pk = 0
for l in k[a:b].splitlines():
    s = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", s) or s.startswith(("s_cbranch", "s_branch")):
        print("%4d insts (%3d v_pk)  then  %s" % (cnt, pk, s[:90])); cnt = pk = 0
    elif s and not s.startswith((";", ".")):
        cnt += 1
        pk += s.startswith("v_pk_")
print("%4d insts (%3d v_pk)  to the end of the section" % (cnt, pk))
