#!/usr/bin/env python3
"""How far do two EAGER runs of the small test loop drift apart (float-atomics order + Adam), and where does the
hipGraph-replayed run sit?  (calibrates the tolerances of tests/test_gpu_graph.py)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_gpu_graph as T  # noqa: E402


def rep(a, b):
    out = {}
    for k in a:
        x, y = a[k].float(), b[k].float()
        out[k.replace("nerf.tensorf.", "")] = "%.1e/%.1e" % (float((x - y).norm() / (x.norm() + 1e-12)),
                                                            float((x - y).abs().max() / (x.abs().max() + 1e-12)))
    return out


for it0 in (0, 9000):
    for K in (6, 12, 24):
        l1, s1, _, _ = T._run(False, K, it0)
        l2, s2, _, _ = T._run(False, K, it0)
        l3, s3, st, _ = T._run(True, K, it0)
        print("it0", it0, "K", K, "loss maxrel: eager-eager %.2e graph-eager %.2e" % (np.abs(l1 / l2 - 1).max(), np.abs(l3 / l1 - 1).max()), st)
        r1, r2 = rep(s1, s2), rep(s1, s3)
        for k in r1:
            if "app_plane.0" in k or "density_plane.0" in k or "mlp.0.weight" in k or "se3" in k:
                print("   %-28s eager-eager %s   graph-eager %s" % (k, r1[k], r2[k]))
