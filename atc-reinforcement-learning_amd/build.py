"""Builds libatcstep.so (the HIP kernels + C-ABI) for gfx950, in-tree, with hipcc.

    python atc-reinforcement-learning_amd/build.py [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off: the integer outputs (done/flags/counters) must match the fp32
oracle bit-for-bit, so the compiler may not fuse a*b+c on its own (explicit fmaf is used where exactness is argued).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "atc_step.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "atc_device.h"), os.path.join(HERE, "csrc", "atc_abi.inc"), os.path.join(HERE, "csrc", "atc_wave.h"),
        os.path.join(HERE, "csrc", "atc_aux_kernels.inc"),
        os.path.join(os.path.dirname(HERE), "include", "atc_step.h")]
OUT = os.path.join(HERE, "atc_hip", "libatcstep.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-kernarg-preload-count=16: the first 64 bytes of the kernel arguments (sector pointer, batch shape, the state record
# pointers) arrive in scalar registers with the wavefront instead of through a first scalar-load round trip (gfx950 supports the
# preload; measured r05: 8 192 x 16 fused 2.34 vs 2.47 us per step, 65 536 x 1 2.98 vs 3.04, the other launches unchanged)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function", "-mllvm", "-amdgpu-kernarg-preload-count=16"]


# Developer micro-benchmark that bench.py runs next to the fused launch (a separate executable, never linked into the product):
# what a launch that only WRITES that launch's outputs takes on the box at hand (tools/ubench/write_bw.hip)
UBENCH_SRC = os.path.join(os.path.dirname(HERE), "tools", "ubench", "write_bw.hip")
UBENCH_OUT = os.path.join(HERE, "atc_hip", "ubench_write_bw")


def build_ubench(force=False, verbose=False):
    if not force and os.path.exists(UBENCH_OUT) and os.path.getmtime(UBENCH_OUT) >= os.path.getmtime(UBENCH_SRC):
        return UBENCH_OUT
    tmp = UBENCH_OUT + ".tmp%d" % os.getpid()
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", tmp, UBENCH_SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, UBENCH_OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return UBENCH_OUT


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return OUT
    extra = list(extra)   # explicit arguments only: nothing in the environment changes what the product is built from
    tmp = OUT + ".tmp%d" % os.getpid()  # link under a private name, publish atomically (other ranks may be waiting)
    cmd = [HIPCC] + FLAGS + extra + ["-o", tmp, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)  # never on stdout: bench.py prints exactly one JSON line there
    try:
        try:
            subprocess.check_call(cmd)
        except subprocess.CalledProcessError:
            # an LLVM that does not know the kernarg-preload option: the same build without it (a 2-5 % slower small-batch launch)
            plain = [c for c in cmd if c not in ("-mllvm", "-amdgpu-kernarg-preload-count=16")]
            if plain == cmd:
                raise
            print("retrying without -amdgpu-kernarg-preload-count", file=sys.stderr)
            subprocess.check_call(plain)
        os.replace(tmp, OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OUT


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv, verbose=True,
              extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ())
    print(OUT)
    print(build_ubench(force="--force" in sys.argv, verbose=True))
