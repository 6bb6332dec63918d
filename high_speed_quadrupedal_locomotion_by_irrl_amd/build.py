"""Build the gfx950 shared library of the env kernels + C-ABI (in-tree, so it travels with gpurun).

    python -m high_speed_quadrupedal_locomotion_by_irrl_amd.build [--force]

hipcc cross-compiles without a GPU; the only target is gfx950 (MI355X / CDNA4).

The env kernels do not go through the compiler driver in one piece: their device code is compiled to assembly, passed
through isa_pass.py (wait states for the hand-placed v_fmac_f32_dpp instructions, which the compiler cannot see through
inline assembly), assembled, linked and bundled by the same LLVM tools the driver would call, and the host half is compiled
with that device binary embedded (`compile_env_unit`).  Kernel registration and launches are the ordinary HIP ones.

Staleness is decided by CONTENT, not by file times: a sha256 over every source under csrc/, include/irrl_env.h and
the compiler flags is baked into the library (`irrl_version()` = "gfx950;irrl-env r3;irrl-src-hash:<hex>"); a
prebuilt `.so` is reused only when the hash found inside it equals the hash of the sources next to it.  (File times
do not survive a fresh checkout or an rsync, and the built `.so` travels to the GPU box although git ignores it.)"""
import hashlib
import os
import re
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(_HERE, "..", "include", "irrl_env.h")
LIB = os.path.join(_HERE, "libirrl_env.so")
_MARK = b"irrl-src-hash:"


def sources(csrc=None):
    """every source / header under csrc/ is a dependency of the library (three translation units include most of them)"""
    csrc = csrc or CSRC
    return sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".h")))


# the reference's own operator boundary, compiled: pybind11 module `_flexible_robot` (csrc/flexible_robot_pybind.cpp) over the
# C-ABI -- a host-only g++ translation unit that links libirrl_env.so, built into <repo>/native/
PYBIND_SRC = os.path.join(CSRC, "flexible_robot_pybind.cpp")
NATIVE_DIR = os.path.join(_HERE, "..", "native")


def pybind_module_path():
    import sysconfig
    return os.path.join(NATIVE_DIR, "_flexible_robot" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind(force=False, verbose=False):
    """-> path of native/_flexible_robot<EXT_SUFFIX>; rebuilt when its source or the C-ABI header is newer (content hash kept next
    to it, like the library's)."""
    import sysconfig
    import pybind11
    out = pybind_module_path()
    h = hashlib.sha256()
    for f in (PYBIND_SRC, HEADER):
        with open(f, "rb") as fh:
            h.update(fh.read())
    # ... and what the binary is tied to besides its sources: the interpreter's ABI tag and pybind11's version
    h.update((sysconfig.get_config_var("EXT_SUFFIX") or "").encode() + b"\0" + getattr(pybind11, "__version__", "").encode())
    want = h.hexdigest()[:16]
    stamp = out + ".srchash"
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return out
    os.makedirs(NATIVE_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           "-I" + os.path.dirname(os.path.abspath(HEADER)), PYBIND_SRC, "-o", out, "-L" + _HERE, "-lirrl_env",
           "-Wl,-rpath,$ORIGIN/../" + os.path.basename(_HERE)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return out


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


# -ffp-contract=on (round 5): a multiply-add fuses where the SOURCE writes `a * b + c` in one expression (the front end emits llvm.fmuladd) and
# nowhere else.  HIP's default, `fast`, lets the back end fuse across statements, and which pairs it picks depends on the code AROUND the
# expression: the same device function inlined into two kernels -- the env step in the step kernel and in the multi-step kernel that carries
# the lane context in registers; the policy step in the stand-alone and in the rollout kernels, which even live in two translation units -- then
# differs in the last bit (measured: 15 of 1250 v_fmac), which breaks the bit-identity the entry points promise.  Same instruction counts and
# the same times within noise for every kernel of the library (env step 41.9 us, LSTM update 101.9 ms, MLP update 18.9 ms).
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fno-signed-zeros", "-ffp-contract=on"]
# Env kernels only (measured on MI355X, 4096 envs, same box): the SLP vectorizer's packed-f32 pairs cost more v_mov than
# they save in this scalar-per-lane code (56.6 -> 51.4 us per step without it), and the single resident wave per SIMD
# wants ILP-first scheduling (-> 50.1 us).
# Round 6: `iterative-ilp` instead of `max-ilp` -- same-box A/B of the five strategies the back end offers on the persistent step kernel: max-ilp 31.7-32.0 us,
# max-memory-clause 31.9, iterative-maxocc 31.6-31.7, the default 31.6, iterative-minreg 33.1, iterative-ilp 31.1-31.5 (profiles/r06_ab_env_sched_strategy_same_box.log).
# Scheduling only: results bit-identical.
ENV_FLAGS = ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
# the 4-lane layout's build for two waves per SIMD (env_kernels.hip: _l4w2, pools of more than 16 384 robots): 256 registers per wave, so the
# strategy that schedules for register pressure first measures best there -- 131 072 envs 483 -> 490 M env-steps/s, 32 768 envs 450 -> 454 M
# (iterative-minreg 488 / 454, the default 484 / 448; profiles/r06_ab_l4_two_waves_per_simd_same_box.log)
ENV_FLAGS_W2 = ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]


def llvm_bin():
    for cand in (os.environ.get("ROCM_LLVM_BIN"), "/opt/rocm/lib/llvm/bin"):
        if cand and os.path.exists(os.path.join(cand, "clang")):
            return cand
    raise RuntimeError("ROCm LLVM tools not found (expected /opt/rocm/lib/llvm/bin)")


def compile_env_unit(src, flags, obj, workdir, verbose=False):
    """env_kernels.hip -> host object with the gfx950 code embedded, through the ISA pass:
    device assembly (hipcc -S) -> isa_pass -> assemble -> link -> offload bundle -> host compile with the bundle included."""
    from . import isa_pass
    tag = os.path.splitext(os.path.basename(obj))[0]
    s_raw, s_fix = os.path.join(workdir, tag + ".raw.s"), os.path.join(workdir, tag + ".s")
    dev_o, dev_out, fatbin = os.path.join(workdir, tag + ".dev.o"), os.path.join(workdir, tag + ".dev.out"), os.path.join(workdir, tag + ".hipfb")
    ll = llvm_bin()
    cmds = [[hipcc()] + flags + ["--cuda-device-only", "-S", src, "-o", s_raw]]
    def run(c):
        """quiet unless it fails: the tool's diagnostics (assembler errors included) are kept and shown with the failure"""
        if verbose:
            print(" ".join(c))
        r = subprocess.run(c, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0 or verbose:
            sys.stderr.write(r.stdout.decode("utf-8", "replace"))
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, c)
    run(cmds[0])
    stats = isa_pass.process_file(s_raw, s_fix)
    if verbose:
        print("[isa_pass] %s: %s" % (tag, stats))
    run([os.path.join(ll, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_fix, "-o", dev_o])
    run([os.path.join(ll, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_out, dev_o])
    run([os.path.join(ll, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + dev_out, "-output=" + fatbin])
    run([hipcc()] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin, "-c", src, "-o", obj])
    return stats


def source_hash(extra_flags=(), csrc=None, header=None):
    """sha256 over (file name, file bytes) of every dependency + the flag lists, first 16 hex digits"""
    csrc = csrc or CSRC
    h = hashlib.sha256()
    with open(os.path.join(_HERE, "isa_pass.py"), "rb") as f:
        h.update(b"isa_pass.py\0" + f.read())
    for name in sources(csrc):
        h.update(name.encode() + b"\0")
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    with open(header or HEADER, "rb") as f:
        h.update(b"irrl_env.h\0" + f.read())
    h.update(("\0".join(COMMON_FLAGS + ["|"] + ENV_FLAGS + ["|"] + ENV_FLAGS_W2 + ["|"] + list(extra_flags))).encode())
    return h.hexdigest()[:16]


def embedded_hash(lib=None):
    """the hash baked into a built library, read from the file's bytes (no dlopen, no HIP runtime); None if absent"""
    lib = lib or LIB
    if not os.path.exists(lib):
        return None
    with open(lib, "rb") as f:
        blob = f.read()
    m = re.search(re.escape(_MARK) + rb"([0-9a-f]{16})", blob)
    return m.group(1).decode() if m else None


def is_stale(extra_flags=(), csrc=None, lib=None):
    return embedded_hash(lib) != source_hash(extra_flags, csrc)


def build(force=False, verbose=False, extra_flags=()):
    """-> path of the library.  Compiles unless the prebuilt one carries the hash of the current sources; prints which."""
    want = source_hash(extra_flags)
    if not force and embedded_hash() == want:
        if verbose:
            print("[build] reusing %s (source hash %s matches)" % (os.path.basename(LIB), want))
        return LIB
    if verbose:
        print("[build] compiling %s (source hash %s, library has %s)" % (os.path.basename(LIB), want, embedded_hash()))
    common = COMMON_FLAGS + list(extra_flags)
    objdir = os.path.join(_HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    # the env kernels in both lane layouts (same source, different lane-primitive header; the 4-lane layout once more for two waves per
    # SIMD) through the ISA pass, then the C-ABI + LSTM kernels through the plain driver
    units = [("env_kernels.hip", ["-DIRRL_LANES_PER_ROBOT=16"] + ENV_FLAGS, "env_kernels_l16.o"),
             ("env_kernels.hip", ["-DIRRL_LANES_PER_ROBOT=4"] + ENV_FLAGS, "env_kernels_l4.o"),
             ("env_kernels.hip", ["-DIRRL_LANES_PER_ROBOT=4", "-DIRRL_L4_WAVES2"] + ENV_FLAGS_W2, "env_kernels_l4w2.o"),
             ("irrl_env_abi.hip", ['-DIRRL_SRC_HASH="%s"' % want], "irrl_env_abi.o")]
    from concurrent.futures import ThreadPoolExecutor
    def one(u):
        src, flags, obj = u
        if src == "env_kernels.hip":
            compile_env_unit(os.path.join(CSRC, src), common + flags, os.path.join(objdir, obj), objdir, verbose)
        else:
            cmd = [hipcc()] + common + flags + ["-c", os.path.join(CSRC, src), "-o", os.path.join(objdir, obj)]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(one, units))
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(objdir, u[2]) for u in units] + ["-o", LIB]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    got = embedded_hash()
    if got != want:
        raise RuntimeError("built library carries source hash %r, expected %r" % (got, want))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_pybind(force="--force" in sys.argv, verbose=True))
