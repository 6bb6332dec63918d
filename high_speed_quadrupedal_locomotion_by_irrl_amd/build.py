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
SOURCES = ["irrl_env_abi.hip", "env_kernels.hip", "lstm_kernels.hip", "env_core.hpp", "lanes_hip.hpp", "env_params.h", "irrl_config.hpp",
           "irrl_state_pool.hpp", "irrl_terrain.hpp"]


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


def build(force=False, verbose=False, extra_flags=()):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-fno-signed-zeros",
           os.path.join(CSRC, "irrl_env_abi.hip"), "-o", LIB] + list(extra_flags)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
