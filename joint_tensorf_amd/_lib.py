"""ctypes binding of libjt_render.so (the C ABI declared in include/jt_render.h).

The library is built in-tree by joint_tensorf_amd/build.py (hipcc --offload-arch=gfx950).  There is
NO fallback: if the shared object is missing or a symbol is absent, importing this module raises.
"""
import ctypes
import os

# torch FIRST: it ships its own libamdhip64 and must be the one that brings the HIP runtime into the process.  Loaded
# the other way round, libjt_render.so pulls /opt/rocm's copy in, torch adds its own, and kernels launched through the
# library see "no device" (hipErrorNoDevice) for memory and streams that belong to the other runtime.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
# JT_LIB_PATH: a differently built copy of the SAME library (kernel-tuning experiments, tools/build_variant.py) -- still the
# HIP path, still no fallback
LIB_PATH = os.environ.get("JT_LIB_PATH") or os.path.join(HERE, "lib", "libjt_render.so")

c_float_p = ctypes.c_void_p  # raw device pointers travel as integers
c_f = ctypes.c_float
c_i = ctypes.c_int32

JT_ACT_SOFTPLUS, JT_ACT_RELU = 0, 1
JT_MLP_FEA, JT_MLP_WEAKVIEW = 0, 1
JT_SHADE_SKIP_WGRAD, JT_SHADE_POSE_ONLY = 1, 2


class JtScene(ctypes.Structure):
    _fields_ = [
        ("aabb_lo", c_f * 3), ("aabb_hi", c_f * 3),
        ("plane_h", c_i * 3), ("plane_w", c_i * 3), ("line_len", c_i * 3),
        ("n_comp_density", c_i), ("n_comp_app", c_i),
        ("step_size", c_f), ("near_plane", c_f), ("far_plane", c_f),
        ("distance_scale", c_f), ("density_shift", c_f), ("density_act", c_i),
        ("weight_thres", c_f), ("n_samples", c_i), ("ndc", c_i), ("white_bg", c_i),
        ("app_dim", c_i), ("mlp_kind", c_i), ("mlp_hidden", c_i), ("view_pe", c_i), ("fea_pe", c_i),
        ("view_pe_progress", c_f), ("fea_pe_progress", c_f),
        ("mask_dims", c_i * 3), ("mask_lo", c_f * 3), ("mask_inv", c_f * 3),
        ("near_plane_dev", ctypes.c_void_p),
    ]


class JtFactors(ctypes.Structure):
    _fields_ = [("density_plane", ctypes.c_void_p * 3), ("density_line", ctypes.c_void_p * 3),
                ("app_plane", ctypes.c_void_p * 3), ("app_line", ctypes.c_void_p * 3),
                ("alpha_volume", ctypes.c_void_p)]


class JtMlp(ctypes.Structure):
    _fields_ = [("basis", ctypes.c_void_p), ("w1", ctypes.c_void_p), ("b1", ctypes.c_void_p),
                ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p), ("w3", ctypes.c_void_p),
                ("b3", ctypes.c_void_p)]


class JtBlurItem(ctypes.Structure):
    _fields_ = [("in_", ctypes.c_void_p), ("out", ctypes.c_void_p), ("tmp", ctypes.c_void_p),
                ("taps", ctypes.c_void_p), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("C", ctypes.c_int32),
                ("n_taps", ctypes.c_int32)]


class JtFiniteItem(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("n", ctypes.c_int64), ("bit", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class JtAdamItem(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_int64), ("lr", ctypes.c_float), ("bias_correction1", ctypes.c_float),
                ("bias_correction2", ctypes.c_float), ("pad_", ctypes.c_int32)]


# every symbol include/jt_render.h declares (tests/test_abi.py checks header <-> this table <-> .so)
P = ctypes.c_void_p
I = ctypes.c_int
F = ctypes.c_float
SP = ctypes.POINTER(JtScene)
FP = ctypes.POINTER(JtFactors)
MP = ctypes.POINTER(JtMlp)
SIGNATURES = {
    "jt_version": (ctypes.c_int, []),
    "jt_chip_geometry": (I, [P]),
    "jt_raygen_forward": (I, [P, P, P, P, I, I, I, I, F, P, P, P]),
    "jt_raygen_backward": (I, [P, P, P, P, I, I, I, I, F, P, P, P, P]),
    "jt_raygen_forward_ragged": (I, [P, P, P, P, P, I, I, I, I, F, P, P, P]),
    "jt_raygen_backward_ragged": (I, [P, P, P, P, P, I, I, I, I, F, P, P, P, P]),
    "jt_render_loss_views_forward": (I, [P, P, P, P, I, I, P, P, P]),
    "jt_render_loss_views_backward": (I, [P, P, P, P, I, I, I, P, P, P, P]),
    "jt_pose_forward": (I, [P, P, P, I, I, P, P]),
    "jt_pose_backward": (I, [P, P, P, I, I, P, P, P]),
    "jt_blur_forward": (I, [P, P, P, I, I, I, P, I, P]),
    "jt_blur_backward": (I, [P, P, P, I, I, I, P, I, P]),
    "jt_lattice_indices": (I, [P, I, I, I, I, P, P]),
    "jt_march_forward": (I, [SP, FP, P, P, P, P, I, P, P, P, P, P, P, P, P, P]),
    "jt_march_forward_pose": (I, [SP, FP, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P]),
    "jt_shade_list": (I, [SP, P, I, P, P, P, P, P, I, P]),
    "jt_composite_forward": (I, [SP, I, P, P, P, P, P, P, P, P]),
    "jt_composite_backward": (I, [SP, I, P, P, P, P, P, P, P, I, P]),
    "jt_march_backward": (I, [SP, FP, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, FP, P, P, P, ctypes.c_size_t, P]),
    "jt_march_backward_pose": (I, [SP, FP, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, ctypes.c_size_t, P]),
    "jt_march_backward_workspace_bytes": (ctypes.c_size_t, [SP, I]),
    "jt_shade_workspace_bytes": (ctypes.c_size_t, [SP, I]),
    "jt_shade_record_layout": (I, [SP, P]),
    "jt_shade_workspace_layout": (I, [SP, I, P]),
    "jt_finite_check": (I, [P, I, P, P]),
    "jt_set_deterministic": (I, [I]),
    "jt_status_bind": (I, [P]),
    "jt_status_clear": (I, [P]),
    "jt_shade_chunk_entries": (I, []),
    "jt_shade_matrix_mode": (I, []),
    "jt_shade_set_matrix_mode": (I, [I]),
    "jt_shade_set_chunk_log2": (I, [I]),
    "jt_shade_bwd_split": (I, []),
    "jt_shade_set_bwd_split": (I, [I]),
    "jt_shade_lean_tape": (I, []),
    "jt_shade_set_lean_tape": (I, [I]),
    "jt_render_loss_forward": (I, [P, P, P, P, I, I, I, F, F, P, P, P]),
    "jt_render_loss_backward": (I, [P, P, P, P, I, I, I, F, F, P, P, P, P]),
    "jt_render_loss_forward_ind": (I, [P, P, P, I, I, I, I, F, F, P, P, P]),
    "jt_render_loss_backward_ind": (I, [P, P, P, I, I, I, I, F, F, P, P, P, P]),
    "jt_tv_depth_forward": (I, [P, I, I, I, P, P]),
    "jt_loss_sum_forward": (I, [P, P, F, F, F, F, P, P]),
    "jt_loss_sum_backward": (I, [P, F, F, F, F, P, P, P]),
    "jt_loss_sum_forward_dyn": (I, [P, P, P, P, P]),
    "jt_loss_sum_check_forward": (I, [P, P, P, P, P, P, I, I, P, P]),
    "jt_loss_sum_backward_dyn": (I, [P, P, P, P, P]),
    "jt_reg_losses_forward": (I, [FP, P, I, I, I, I, P, P, P]),
    "jt_reg_losses_backward": (I, [FP, P, I, I, P, I, I, FP, I, P, P]),
    "jt_reg_losses_fused": (I, [FP, P, I, I, I, I, P, P, FP, P, P, P]),
    "jt_adam_step": (I, [P, I, F, F, F, P]),
    "jt_adam_step_dyn": (I, [P, I, F, F, F, P, P]),
    "jt_adam_step_coefs": (I, [P, I, F, F, F, P, P]),
    "jt_poke": (I, [P, P, I, P]),
    "jt_dense_alpha": (I, [SP, FP, P, ctypes.c_long, F, P, P]),
    "jt_blur_batch_forward": (I, [P, I, P]),
    "jt_blur_batch_backward": (I, [P, I, P]),
    "jt_factor_reg_forward": (I, [P, I, I, I, P, P]),
    "jt_factor_reg_backward": (I, [P, I, I, I, P, P, I, P]),
    "jt_shade_forward": (I, [SP, FP, MP, P, P, P, P, P, P, I, P, P, P, P, I, P, ctypes.c_size_t, I, P]),
    "jt_shade_backward": (I, [SP, FP, MP, P, P, P, P, P, P, I, P, P, P, P, P, FP, MP, P, I, P, ctypes.c_size_t, I, P, P, P, P]),
}


# the OPTIONAL module include/jt_fused.h declares (libjt_fused.so: the single-launch test-time kernel, loaded on demand)
FUSED_SIGNATURES = {
    "jt_pose_fused_workspace_bytes": (ctypes.c_size_t, [SP]),
    "jt_pose_fused": (I, [SP, FP, MP, P, P, P, I, P, P, I, I, F, P, P, P, P, P, P, P, P, ctypes.c_size_t, P]),
}
FUSED_LIB_PATH = os.path.join(HERE, "lib", "libjt_fused.so")
_FUSED = []


class JtError(RuntimeError):
    pass


def fused_lib():
    """libjt_fused.so, loaded the first time somebody asks for the single-launch kernel (opt.optim.test_fused).  A library
    variant that still carries the two symbols itself (tools/build_variant.py builds every source into one object) serves them."""
    if not _FUSED:
        src = lib if hasattr(lib, "jt_pose_fused") else None
        if src is None:
            if not os.path.exists(FUSED_LIB_PATH):
                raise ImportError("joint_tensorf_amd: %s is missing -- `python joint_tensorf_amd/build.py` builds it" % FUSED_LIB_PATH)
            src = ctypes.CDLL(FUSED_LIB_PATH)
        for name, (res, args) in FUSED_SIGNATURES.items():
            fn = getattr(src, name)
            fn.restype = res
            fn.argtypes = args
        _FUSED.append(src)
    return _FUSED[0]


# the JT_VERSION of include/jt_render.h that SIGNATURES and the struct mirrors above were written against.  A constant, not a
# read of the header at import time: a vendored copy of the package has no include/ directory beside it (tests/test_abi.py
# holds this number, the header's and the library's together)
JT_ABI_VERSION = 1204


def header_version():
    """JT_VERSION of include/jt_render.h (tests only: the repo's include/ directory has to exist)"""
    import re
    src = open(os.path.join(os.path.dirname(HERE), "include", "jt_render.h")).read()
    return int(re.search(r"^#define\s+JT_VERSION\s+(\d+)", src, flags=re.M).group(1))


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "joint_tensorf_amd: %s is missing -- build it with `python joint_tensorf_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    # a library built from another revision of the header (a stale .so, a JT_LIB_PATH variant of an older tree) would be
    # handed structs and scratch buffers of the wrong size: refuse it here instead of inside a kernel
    want, got = JT_ABI_VERSION, lib.jt_version()
    if want != got:
        raise ImportError("joint_tensorf_amd: %s reports ABI version %d, this binding was written against %d -- rebuild with "
                          "`python joint_tensorf_amd/build.py --force`" % (LIB_PATH, got, want))
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        kind = "bad argument" if rc == 1 else "unsupported shape" if rc == 2 else "HIP error %d" % (-rc)
        raise JtError("%s failed: %s (rc=%d)" % (what, kind, rc))


def ptr(t):
    """device pointer of a tensor (or None).  A host tensor here would fault inside a kernel: refuse it."""
    if t is None:
        return None
    if not t.is_cuda:
        raise JtError("host tensor handed to a device entry point (shape %s)" % (tuple(t.shape),))
    return t.data_ptr()
