"""Build the gfx950 shared library of the env kernels + C-ABI (in-tree, so it travels with gpurun).

    python -m high_speed_quadrupedal_locomotion_by_irrl_amd.build

hipcc cross-compiles without a GPU; the only target is gfx950 (MI355X / CDNA4)."""
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libirrl_env.so")
# every source / header under csrc/ is a dependency of the library (three translation units include most of them)
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(_HERE, "..", "include", "irrl_env.h")]
    return os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps)


# Env kernels only (measured on MI355X, 4096 envs, same box): the SLP vectorizer's packed-f32 pairs cost more v_mov than
# they save in this scalar-per-lane code (56.6 -> 51.4 us per step without it), and the single resident wave per SIMD
# wants ILP-first scheduling (-> 50.1 us).
ENV_FLAGS = ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def build(force=False, verbose=False, extra_flags=()):
    if not force and not is_stale():
        return LIB
    common = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fno-signed-zeros"] + list(extra_flags)
    objdir = os.path.join(_HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    # the env kernels in both lane layouts (same source, different lane-primitive header), then the C-ABI + LSTM kernels
    units = [("env_kernels.hip", ["-DIRRL_LANES_PER_ROBOT=16"] + ENV_FLAGS, "env_kernels_l16.o"),
             ("env_kernels.hip", ["-DIRRL_LANES_PER_ROBOT=4"] + ENV_FLAGS, "env_kernels_l4.o"),
             ("irrl_env_abi.hip", [], "irrl_env_abi.o")]
    procs = []
    for src, flags, obj in units:
        cmd = common + flags + ["-c", os.path.join(CSRC, src), "-o", os.path.join(objdir, obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(objdir, u[2]) for u in units] + ["-o", LIB]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
