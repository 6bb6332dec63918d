"""Build A/B variants of libirrl_env.so (same ABI, different -D switches) into csrc/_variants/ for one-call GPU comparisons.
usage: python tools/build_variants.py name=-DFLAG[,-DFLAG2] ...   (name 'base' = no extra flags)"""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from high_speed_quadrupedal_locomotion_by_irrl_amd import build as B

out = os.path.join(B.CSRC, "_variants")
os.makedirs(out, exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    flags = [f for f in flags.split(",") if f]
    objs = []
    # "envsched:<strategy>" replaces the env kernels' scheduling strategy (build.ENV_FLAGS: iterative-ilp); "envnosched" drops it
    env_flags = list(B.ENV_FLAGS)
    swapped = any(f.startswith("envsched:") or f == "envnosched" for f in flags)    # (then the two-waves unit takes the same strategy as the others)
    for f in list(flags):
        if f.startswith("envsched:"):
            flags.remove(f)
            env_flags = [x for x in env_flags if not x.startswith("-amdgpu-sched-strategy")]
            env_flags[env_flags.index("-mllvm") + 1:env_flags.index("-mllvm") + 1] = ["-amdgpu-sched-strategy=" + f.split(":", 1)[1]]
        elif f == "envnosched":
            flags.remove(f)
            i = env_flags.index("-mllvm")
            del env_flags[i:i + 2]
    B_ENV = env_flags
    common = list(B.COMMON_FLAGS) + flags
    for src, fl, obj in (("env_kernels.hip", ["-DIRRL_LANES_PER_ROBOT=16"] + B_ENV, "l16"), ("env_kernels.hip", ["-DIRRL_LANES_PER_ROBOT=4"] + B_ENV, "l4"),
                         ("env_kernels.hip", ["-DIRRL_LANES_PER_ROBOT=4", "-DIRRL_L4_WAVES2"] + (B_ENV if swapped else list(B.ENV_FLAGS_W2)), "l4w2"),
                         ("irrl_env_abi.hip", [], "abi")):
        o = os.path.join(out, f"{name}_{obj}.o")
        objs.append(o)
        if src == "env_kernels.hip":   # same route as the product build: device assembly -> ISA pass -> assemble -> embed
            B.compile_env_unit(os.path.join(B.CSRC, src), common + fl, o, out)
        else:
            subprocess.check_call([B.hipcc()] + common + fl + ["-c", os.path.join(B.CSRC, src), "-o", o])
    lib = os.path.join(out, f"libirrl_env_{name}.so")
    subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    for o in objs:
        os.remove(o)
    for f in os.listdir(out):     # the ISA route's intermediates (they would travel to the GPU box with every gpurun snapshot)
        if f.startswith(name + "_l") and f.split(".", 1)[-1] in ("dev.o", "dev.out", "hipfb", "raw.s", "s"):
            os.remove(os.path.join(out, f))
    print(lib)
