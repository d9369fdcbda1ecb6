#!/usr/bin/env python3
"""Build a variant of libjt_render.so with extra compiler flags (kernel-tuning experiments on the GPU box):

  python tools/build_variant.py <name> -DJT_WALK_WAVES=7 ...   ->  joint_tensorf_amd/lib/variants/<name>.so

Run anything against it with JT_LIB_PATH=joint_tensorf_amd/lib/variants/<name>.so (relative to the repo root on the GPU
box: $GRAFT_REPO_ROOT/...).  The variants are git-ignored build artefacts; they travel with gpurun snapshots."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "joint_tensorf_amd")


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(PKG, "lib", "variants")
    obj_dir = os.path.join(out_dir, name + "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))

    def cc(src):
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
                               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "csrc"), "-Wno-unused-function"]
                              + flags + ["-c", src, "-o", obj])
        return obj
    with ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(cc, srcs))
    lib = os.path.join(out_dir, name + ".so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
