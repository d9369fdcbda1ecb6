"""Does anything that shares a CU with the backward's kernels change their result?  (Round 5: with re-tiled weight-gradient GEMMs
co-resident beside k_shade_scatter one full-size parity run showed app_line.1 off by 3.5e-4 of its maximum.)  The full-size
bat_blender_VM / bat_llff_VM_MLP iteration of tests/test_gpu_fullsize.py, run quietly and with a side stream kept busy with small
kernels that fit beside a scatter workgroup (stock elementwise kernels: no LDS, few registers; reductions: a few KB of LDS);
prints, per gradient tensor, max |diff| / max |ref| between two quiet runs (the float-atomics noise) and between a quiet and a
stressed run.
  python tools/round5/stress_scatter.py [bat_blender_VM|bat_llff_VM_MLP] [repeats]"""
import sys
import os
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import fullsize_util as U

cfg = sys.argv[1] if len(sys.argv) > 1 else "bat_blender_VM"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kw = dict(density_scale=25.0) if cfg == "bat_blender_VM" else {}
opt, model, var, it0 = U.build(cfg, stage=-1, **kw)
run_kw = {} if cfg == "bat_blender_VM" else dict(offsets=(2, 3), coin=0.7)
side = torch.cuda.Stream()
noise_a = torch.ones(64 * 1024 * 1024, device="cuda")
noise_b = torch.ones(16 * 1024 * 1024, device="cuda")


def stress(kind, n):
    with torch.cuda.stream(side):
        for _ in range(n):
            if kind == "elementwise":
                noise_a.mul_(1.0000001)
            else:
                noise_b.sum()


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


ref = U.run_hip(opt, model, var, **run_kw)["grads"]
worst = {}
for mode in ["quiet"] + ["elementwise", "reduction"] * reps:
    if mode != "quiet":
        torch.cuda.synchronize()
        stress(mode, 6000)   # ~0.5 s of side-stream work: the whole iteration runs beside it
    t0 = time.perf_counter()
    g = U.run_hip(opt, model, var, **run_kw)["grads"]   # (ends with a device-wide synchronisation: waits for the side stream too)
    busy = "%.0f ms" % ((time.perf_counter() - t0) * 1e3)
    for k in g:
        e = relmax(g[k], ref[k])
        key = (mode, k)
        worst[key] = max(worst.get(key, 0.0), e)
    print(mode, "iteration + wait for the side stream:", busy, " worst tensor: %s %.2e"
          % max(((k, relmax(g[k], ref[k])) for k in g), key=lambda t: t[1]), flush=True)
    torch.cuda.synchronize()
print("%-22s %10s %12s %10s" % ("tensor", "quiet", "elementwise", "reduction"))
for k in ref:
    print("%-22s %10.2e %12.2e %10.2e" % (k, worst.get(("quiet", k), 0), worst.get(("elementwise", k), 0), worst.get(("reduction", k), 0)))
