"""The two builds of the matrix stages -- fp32 matrix cores (mode 0) and bf16 matrix cores with three-piece operands
(mode 3, the default) -- rendered on identical state in one process: colours, ray gradients, factor gradients and weight
gradients of the two agree to fp32 rounding level, far inside the tolerance either is held to against the oracle
(tests/test_gpu_parity.py, test_gpu_fullsize.py).  The reference has ONE fp32 chain (tensorBase.py:43-131); a matrix mode is
a scheduling decision of this build and must not be visible in the numbers."""
import numpy as np
import pytest
import torch

from tests.test_gpu_edge import _batch
from tests.test_gpu_fuzz import _scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _render(tf, o, d, S, jit, cot):
    og, dg = o.to(DEV).requires_grad_(True), d.to(DEV).requires_grad_(True)
    for p in tf.parameters():
        p.grad = None
    tf.jitter_override = jit.to(DEV)
    tf.coin_override = 0.9
    try:
        out = tf(None, og, dg, white_bg=True, is_train=True, ndc_ray=False, N_samples=S, view_pe_progress=0.7,
                 fea_pe_progress=0.4)
    finally:
        tf.jitter_override = None
        tf.coin_override = None
    ((out[0] * cot[0]).sum() + (out[2] * cot[1]).sum()).backward()
    grads = {n: p.grad.detach().double().cpu().numpy() for n, p in tf.named_parameters() if p.grad is not None}
    return (out[0].detach().double().cpu().numpy(), out[2].detach().double().cpu().numpy(),
            og.grad.double().cpu().numpy(), dg.grad.double().cpu().numpy(), grads)


@pytest.mark.parametrize("kind", ["blender", "llff"])
def test_bf16x3_stages_match_the_fp32_matrix_core_stages(kind):
    from joint_tensorf_amd import _lib
    lib = _lib.lib
    tf, cfg, params, rs = _scene(31 + (kind == "llff"), kind)
    S = 96
    o, d = _batch([("hit", 300), ("graze", 20), ("miss", 4)], seed=5)
    R = o.shape[0]
    jit = torch.rand(R, 1, generator=torch.Generator().manual_seed(11))
    gc = torch.Generator().manual_seed(13)
    cot = [torch.randn(R, 3, generator=gc).to(DEV), torch.randn(R, generator=gc).to(DEV)]
    prev = lib.jt_shade_set_matrix_mode(-1)
    det = lib.jt_set_deterministic(1)  # fixed-order sums: what differs between the two runs is the matrix stages alone
    try:
        res = {}
        for mode in (0, 7):   # 7: forward chain, weight gradients and (where the backward runs split: the LLFF kind) its chain
            assert lib.jt_shade_set_matrix_mode(mode) in range(8)
            assert lib.jt_shade_matrix_mode() == mode
            res[mode] = _render(tf, o, d, S, jit, cot)
        lib.jt_shade_set_matrix_mode(9)  # out of range: a query
        assert lib.jt_shade_matrix_mode() == 7
    finally:
        lib.jt_set_deterministic(det)
        lib.jt_shade_set_matrix_mode(prev)
    a, b = res[0], res[7]
    assert np.abs(a[0] - b[0]).max() < 2e-6, "colours"          # fp32 rounding of O(1) colours
    assert np.abs(a[1] - b[1]).max() < 2e-5 * max(1.0, np.abs(a[1]).max()), "depth"

    def rel(x, y):
        return np.abs(x - y).max() / max(np.abs(x).max(), 1e-30)
    assert rel(a[2], b[2]) < 2e-5 and rel(a[3], b[3]) < 2e-5, "ray gradients"
    assert set(a[4]) == set(b[4]) and len(a[4]) >= 19
    worst = max((rel(a[4][n], b[4][n]), n) for n in a[4])
    assert worst[0] < 2e-5, "gradient of %s differs by %.2e of its maximum" % (worst[1], worst[0])
    # and the modes are not the same code: some bit of some output differs
    assert any(np.any(a[i] != b[i]) for i in range(4)) or any(np.any(a[4][n] != b[4][n]) for n in a[4])
