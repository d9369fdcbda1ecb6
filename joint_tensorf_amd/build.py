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
SRCS = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
HDRS = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + [os.path.join(ROOT, "include", "jt_render.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SRCS + HDRS + [os.path.abspath(__file__)])


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
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
