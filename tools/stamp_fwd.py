#!/usr/bin/env python3
"""Where a wave of k_shade_fwd_b16 spends its cycles, phase by phase (profiling build:
   python tools/build_variant.py stamp -DJT_STAMP=1; JT_LIB_PATH=joint_tensorf_amd/lib/variants/stamp.so python tools/stamp_fwd.py)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    from joint_tensorf_amd import _lib
    sys.argv = ["bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-probe", "--no-torch-baseline", "--no-extras"]
    import runpy
    lib = _lib.lib
    fn = lib.jt_debug_read_stamps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    try:
        runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
    except SystemExit:
        pass
    out = (ctypes.c_ulonglong * 8)()
    assert fn(out) == 0
    tiles = max(int(out[5]), 1)
    names = ["gather + products + basis product", "encodings + layer 1", "ReLU words + H1 records", "layer 2",
             "MID records + layer 3 + colours"]
    tot = sum(int(out[i]) for i in range(5))
    print("tiles %d, cycles per tile and wave (s_memtime ticks): total %.0f" % (tiles, tot / tiles))
    for i, n in enumerate(names):
        print("  %-36s %8.0f  %5.1f %%" % (n, int(out[i]) / tiles, 100.0 * int(out[i]) / tot))


if __name__ == "__main__":
    main()
