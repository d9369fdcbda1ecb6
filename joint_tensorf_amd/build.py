"""Build the HIP library for gfx950 (MI355X) in-tree: joint_tensorf_amd/lib/libjt_render.so.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib", "libjt_render.so")
# the single-launch test-time kernel is an OPTIONAL module of its own (include/jt_fused.h): slower than the staged kernels, used
# by no default path, and 0.8 MB of code objects that the product library does not carry any more
FUSED_SRC = os.path.join(HERE, "csrc", "jt_fused.hip")
FUSED_LIB = os.path.join(HERE, "lib", "libjt_fused.so")
SRCS = [p for p in sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip"))) if p != FUSED_SRC]
HDRS = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + [os.path.join(ROOT, "include", "jt_render.h"),
                                                               os.path.join(ROOT, "include", "jt_fused.h")]
# test infrastructure, NOT part of the product library: the staged appearance path's two entry points
# (tests/csrc/jt_app.hip -> tests/lib/libjt_test_staged.so), loaded by tests/staged_path.py only
TEST_LIB = os.path.join(ROOT, "tests", "lib", "libjt_test_staged.so")
TEST_SRCS = sorted(glob.glob(os.path.join(ROOT, "tests", "csrc", "*.hip")))
TEST_HDRS = sorted(glob.glob(os.path.join(ROOT, "tests", "csrc", "*.h")))


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(FUSED_LIB) or (TEST_SRCS and not os.path.exists(TEST_LIB)):
        return True
    if any(os.path.getmtime(p) > os.path.getmtime(FUSED_LIB) for p in [FUSED_SRC] + HDRS):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(p) > t for p in SRCS + HDRS + [os.path.abspath(__file__)]):
        return True
    t = os.path.getmtime(TEST_LIB) if TEST_SRCS else 0
    return any(os.path.getmtime(p) > t for p in TEST_SRCS + TEST_HDRS + HDRS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    for src in SRCS:
        obj = os.path.join(HERE, "lib", os.path.basename(src) + ".o")
        if (not force and os.path.exists(obj)
                and all(os.path.getmtime(obj) > os.path.getmtime(p) for p in [src] + HDRS + [os.path.abspath(__file__)])):
            objs.append(obj)
            continue
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"),
               "-Wall", "-Wno-unused-function", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if force or not os.path.exists(FUSED_LIB) or any(os.path.getmtime(p) > os.path.getmtime(FUSED_LIB)
                                                     for p in [FUSED_SRC] + HDRS + [os.path.abspath(__file__)]):
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-shared",
               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-Wall", "-Wno-unused-function",
               "-o", FUSED_LIB, FUSED_SRC]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if TEST_SRCS:
        os.makedirs(os.path.dirname(TEST_LIB), exist_ok=True)
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-shared",
               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-I", os.path.join(ROOT, "tests", "csrc"),
               "-Wall", "-Wno-unused-function", "-o", TEST_LIB] + TEST_SRCS
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


SAN_LIB = os.path.join(ROOT, "tests", "lib", "libjt_render_asan.so")


def asan_runtime():
    """the shared AddressSanitizer runtime of the toolchain that builds the library (preloaded into the python that loads SAN_LIB)"""
    clang = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    return subprocess.check_output([clang, "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()


def build_sanitized(force=False, verbose=False):
    """The HOST side of libjt_render.so under AddressSanitizer + UndefinedBehaviorSanitizer (test infrastructure:
    tests/test_sanitizers.py; never shipped, never run on a GPU box -- GPU-side sanitizers are not available on this pool).
    Device code is compiled as usual but unoptimised (-Xarch_device -O1): it only has to link."""
    if not force and os.path.exists(SAN_LIB) and all(os.path.getmtime(SAN_LIB) > os.path.getmtime(p)
                                                      for p in SRCS + HDRS + [os.path.abspath(__file__)]):
        return SAN_LIB
    import concurrent.futures
    out_dir = os.path.join(ROOT, "tests", "lib", "asan_obj")
    os.makedirs(out_dir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize", "-Xarch_device", "-O1", "-Xarch_device", "-g0",
             "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc")]

    def one(src):
        obj = os.path.join(out_dir, os.path.basename(src) + ".o")
        cmd = [hipcc()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, SRCS))
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan",
                           "-o", SAN_LIB] + objs)
    return SAN_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
