#!/usr/bin/env python3
"""Where a wave of k_tile_scatter spends its time, phase by phase (profiling build:
   python tools/build_variant.py tstamp -DJT_TILE_STAMP=1
   JT_LIB_PATH=joint_tensorf_amd/lib/variants/tstamp.so JT_BWD_SPLIT=1 JT_TILE_CFG=1 python tools/round5/stamp_tile.py)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    from joint_tensorf_amd import _lib
    sys.argv = ["bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-probe", "--no-torch-baseline", "--no-extras"]
    import runpy
    lib = _lib.lib
    fn = lib.jt_debug_read_tile_stamps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    try:
        runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
    except SystemExit:
        pass
    out = (ctypes.c_ulonglong * 16)()
    assert fn(out) == 0
    groups = max(int(out[15]), 1)
    names = ["prefetch issue, tile switch", "pair geometry, records out, W / Wx / Wy", "records back, W^T", "line taps issued",
             "matrix products + per-channel math + line atomics", "coordinate-gradient sums and stores", "slice flush (last group)",
             "rotation, next group", "batch grab", "batch set-up (descriptors, first loads)"]
    tot = sum(int(out[i]) for i in range(10))
    print("groups %d, s_memtime ticks per group and wave: total %.0f" % (groups, tot / groups))
    for i, n in enumerate(names):
        print("  %-52s %8.1f  %5.1f %%" % (n, int(out[i]) / groups, 100.0 * int(out[i]) / tot))


if __name__ == "__main__":
    main()
