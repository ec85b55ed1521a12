"""Build libaqua_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The shared object lands in aquaticgymenv_amd/lib/ so that it travels with the source tree to the
GPU box (a JIT cache under $HOME would not).  hipcc cross-compiles for gfx950 without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "aqua_hip.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "aqua_device.hpp"), os.path.join(HERE, "csrc", "aqua_tuning.inc"),
              os.path.join(os.path.dirname(HERE), "include", "aqua_hip.h")]
LIB = os.path.join(HERE, "lib", "libaqua_hip.so")
ARCH = "gfx950"
# -fno-honor-nans: no state ever holds a NaN; it drops the v_max(x, x) canonicalisation in front of every
# fmin/fmax (results for non-NaN inputs are unchanged; contraction is controlled per function by pragmas)
# -cuid=...: clang derives the compilation-unit id it bakes into the fat binary's symbols from the source's absolute path
# by default -- the same source built in another directory is then another file.  Fixed, the library's hash (the build
# tag of bench.py and profiles/traffic.json) depends on the sources and this command line only.
COMMON_FLAGS = ["-O3", "--offload-arch=" + ARCH, "-std=c++17", "-shared", "-fPIC", "-fno-honor-nans", "-cuid=aqua_hip"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_hip(force=False, verbose=False, extra_flags=()):
    """Compile csrc/aqua_hip.hip -> lib/libaqua_hip.so for gfx950.  Returns the library path."""
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc_path(), *COMMON_FLAGS, "-Wall", "-Wno-unused-function", *extra_flags, "-o", LIB + ".tmp", *SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_variant(name, flags, verbose=False):
    """Tuning builds for A/B timing (tools/ablate.py): lib/variants/libaqua_hip_<name>.so with extra -D flags.
    Select one at run time with AQUA_HIP_LIB=<path>."""
    out = os.path.join(HERE, "lib", "variants", "libaqua_hip_%s.so" % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc_path(), *COMMON_FLAGS, *flags, "-o", out, *SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--variants" in sys.argv:
        for name, flags in (("stamps", ["-DAQUA_STAMPS=1"]), ("nw", ["-DAQUA_NS_NOWORK"]), ("nm", ["-DAQUA_NS_NOMAIN"])):
            print(build_variant(name, flags, verbose=True))
    print(build_hip(force="--force" in sys.argv, verbose=True))
