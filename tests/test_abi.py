"""The C-ABI library loads and exports exactly the symbols include/jt_render.h declares, and the ctypes
table in joint_tensorf_amd/_lib.py covers them with matching argument counts (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "jt_render.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t)\s+(jt_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(2)] = n
    return out


def test_header_vs_ctypes_vs_library():
    funcs = _header_functions()
    assert len(funcs) >= 18, funcs
    from joint_tensorf_amd import _lib
    assert set(funcs) == set(_lib.SIGNATURES), set(funcs) ^ set(_lib.SIGNATURES)
    for name, nargs in funcs.items():
        assert len(_lib.SIGNATURES[name][1]) == nargs, (name, nargs, len(_lib.SIGNATURES[name][1]))
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in funcs:
        assert hasattr(so, name), name
    assert so.jt_version() >= 1001


def test_struct_layout_matches_header():
    """JtScene is 4-byte fields only; its ctypes mirror must have the same size as the C struct
    (3+3 floats, 9 ints, 2 ints, 5 floats, 1 int, 1 float, 8 ints, 2 floats = 34 words)."""
    from joint_tensorf_amd import _lib
    assert ctypes.sizeof(_lib.JtScene) == 34 * 4
    assert ctypes.sizeof(_lib.JtFactors) == 12 * 8
    assert ctypes.sizeof(_lib.JtMlp) == 7 * 8


def test_bad_arguments_are_rejected_without_a_gpu():
    from joint_tensorf_amd import _lib
    assert _lib.lib.jt_pose_forward(None, None, None, 12, 4, None, None) == 1
    assert _lib.lib.jt_blur_forward(None, None, None, 4, 4, 16, None, 65, None) == 1
    s = _lib.JtScene()
    assert _lib.lib.jt_shade_workspace_bytes(ctypes.byref(s), 1000) == 0  # unsupported shape -> 0
