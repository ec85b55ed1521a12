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


class NoSanitizerRuntime(RuntimeError):
    pass


def build_hostsan(verbose=False):
    """lib/variants/libaqua_hip_hostsan.so: the HOST half of the library (argument validation, the blob / table packers,
    graph / event / IPC handle bookkeeping) instrumented with AddressSanitizer + UBSan; device code untouched
    (-fno-gpu-sanitize: no instrumented kernels, no xnack code objects -- GPU ASan does not exist on this pool and is never
    asked for).  CPU box only: tests/test_sanitizers.py runs tests/test_capi_cpu.py against it in a child process with the
    runtime preloaded.  -> (library path, path of clang's shared ASan runtime to preload)."""
    cc = hipcc_path()
    clang = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(cc))), "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = shutil.which("amdclang") or shutil.which("clang") or ""
    runtime = ""
    if clang:
        runtime = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
        if not os.path.isabs(runtime):           # newer layouts: lib/<triple>/libclang_rt.asan.so
            runtime = subprocess.run([clang, "-print-file-name=libclang_rt.asan.so"], capture_output=True, text=True).stdout.strip()
    if not runtime or not os.path.isabs(runtime) or not os.path.exists(runtime):
        raise NoSanitizerRuntime("hipcc's clang has no shared AddressSanitizer runtime (looked for libclang_rt.asan-x86_64.so)")
    out = os.path.join(HERE, "lib", "variants", "libaqua_hip_hostsan.so")
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in DEPS + [os.path.abspath(__file__)]):
        return out, runtime
    # -DAQUA_DEV_U8_ONLY: the device side is compiled for one action kind instead of seven (a sixth of the compile time);
    # the host code under test -- validation, packers, handles -- is the same, only the launch tables are shorter
    flags = ["-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
             "-fno-gpu-sanitize", "-shared-libsan", "-DAQUA_DEV_U8_ONLY"]
    return build_variant("hostsan", flags, verbose=verbose), runtime


if __name__ == "__main__":
    if "--variants" in sys.argv:
        for name, flags in (("stamps", ["-DAQUA_STAMPS=1"]), ("nw", ["-DAQUA_NS_NOWORK"]), ("nm", ["-DAQUA_NS_NOMAIN"])):
            print(build_variant(name, flags, verbose=True))
    print(build_hip(force="--force" in sys.argv, verbose=True))
