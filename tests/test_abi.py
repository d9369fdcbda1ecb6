"""The C-ABI library loads and exports exactly the symbols include/jt_render.h declares, and the ctypes
table in joint_tensorf_amd/_lib.py covers them with matching argument counts (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(name="jt_render.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
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
    # the library and the header agree on the ABI revision (a mismatch is an ImportError in _lib._load as well)
    assert so.jt_version() == _lib.header_version() == _lib.JT_ABI_VERSION >= 1200


def abi_fingerprint(name="jt_render.h"):
    """sha256 over what a caller compiles against: the header with comments and white space removed (prototypes, struct
    members in order, constants).  Comment edits do not move it; anything a caller could observe does."""
    import hashlib
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#define\s+JT_VERSION\s+\d+", "", src)
    return hashlib.sha256(re.sub(r"\s+", "", src).encode()).hexdigest()


def test_abi_hash_moves_only_with_the_version():
    """include/jt_render.abi = "<JT_VERSION> <fingerprint>" as of the last deliberate ABI change.  A prototype, struct or
    constant edited without a JT_VERSION bump fails here (three such commits the day the number was
    introduced); after a bump, `python tests/test_abi.py --update` rewrites the file."""
    from joint_tensorf_amd import _lib
    ver, fp = open(os.path.join(ROOT, "include", "jt_render.abi")).read().split()
    cur = abi_fingerprint()
    if cur != fp:
        assert _lib.header_version() != int(ver), (
            "include/jt_render.h changed what a caller sees but JT_VERSION is still %s: bump it (and _lib.JT_ABI_VERSION), "
            "then run `python tests/test_abi.py --update`" % ver)
        raise AssertionError("JT_VERSION was bumped to %d: run `python tests/test_abi.py --update` to record the new "
                             "fingerprint" % _lib.header_version())
    assert int(ver) == _lib.header_version()


def test_optional_fused_module():
    """include/jt_fused.h <-> _lib.FUSED_SIGNATURES <-> libjt_fused.so; and the product library no longer carries the
    single-launch kernel (slower than what it would replace -- out of libjt_render.so)."""
    funcs = _header_functions("jt_fused.h")
    from joint_tensorf_amd import _lib
    assert set(funcs) == set(_lib.FUSED_SIGNATURES) == {"jt_pose_fused", "jt_pose_fused_workspace_bytes"}
    for name, nargs in funcs.items():
        assert len(_lib.FUSED_SIGNATURES[name][1]) == nargs, (name, nargs)
    so = ctypes.CDLL(_lib.FUSED_LIB_PATH)
    main = ctypes.CDLL(_lib.LIB_PATH)
    for name in funcs:
        assert hasattr(so, name) and name not in _lib.SIGNATURES and name not in _header_functions()
        if "JT_LIB_PATH" not in os.environ:
            assert not hasattr(main, name), name


def test_product_library_has_no_test_only_entry_points():
    """the staged cross-check path's kernels (tests/csrc/jt_app.hip) are built into tests/lib/libjt_test_staged.so, not
    into the product library or its header (hygiene)"""
    from joint_tensorf_amd import _lib
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in ("jt_app_gather_forward", "jt_app_gather_backward"):
        assert not hasattr(so, name) and name not in _lib.SIGNATURES and name not in _header_functions()
    t = ctypes.CDLL(os.path.join(ROOT, "tests", "lib", "libjt_test_staged.so"))
    assert hasattr(t, "jt_app_gather_forward") and hasattr(t, "jt_app_gather_backward")


def test_struct_layout_matches_header(tmp_path):
    """The ctypes mirrors must have the size and field offsets the C compiler gives the structs of the header."""
    import os
    import subprocess
    from joint_tensorf_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "jt_render.h"\n'
        "int main(void) {\n"
        '  printf("%zu %zu %zu %zu\\n", sizeof(JtScene), sizeof(JtFactors), sizeof(JtMlp), sizeof(JtBlurItem));\n'
        '  printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(JtScene, n_comp_density), offsetof(JtScene, fea_pe_progress),\n'
        "         offsetof(JtScene, mask_dims), offsetof(JtScene, mask_inv), offsetof(JtFactors, alpha_volume),\n"
        "         offsetof(JtScene, near_plane_dev));\n"
        "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    sizes, offs = [list(map(int, l.split())) for l in subprocess.check_output([str(exe)]).decode().splitlines()]
    assert sizes == [ctypes.sizeof(_lib.JtScene), ctypes.sizeof(_lib.JtFactors), ctypes.sizeof(_lib.JtMlp),
                     ctypes.sizeof(_lib.JtBlurItem)]
    assert offs == [_lib.JtScene.n_comp_density.offset, _lib.JtScene.fea_pe_progress.offset,
                    _lib.JtScene.mask_dims.offset, _lib.JtScene.mask_inv.offset, _lib.JtFactors.alpha_volume.offset,
                    _lib.JtScene.near_plane_dev.offset]
    assert ctypes.sizeof(_lib.JtScene) == 43 * 4 + 4 + 8  # 43 four-byte fields, padding, near_plane_dev


def test_bad_arguments_are_rejected_without_a_gpu():
    from joint_tensorf_amd import _lib
    assert _lib.lib.jt_pose_forward(None, None, None, 12, 4, None, None) == 1
    assert _lib.lib.jt_blur_forward(None, None, None, 4, 4, 16, None, 65, None) == 1
    s = _lib.JtScene()
    assert _lib.lib.jt_shade_workspace_bytes(ctypes.byref(s), 1000) == 0  # unsupported shape -> 0
    # the device-memory variants used under hipGraph replay insist on their device arrays
    assert _lib.lib.jt_render_loss_forward_ind(None, None, None, 0, 1, 1, 1, 1.0, 1.0, None, None, None) == 1
    assert _lib.lib.jt_render_loss_backward_ind(None, None, None, 0, 1, 1, 1, 1.0, 1.0, None, None, None, None) == 1
    assert _lib.lib.jt_loss_sum_forward_dyn(None, None, None, None, None) == 1
    assert _lib.lib.jt_finite_check(None, 0, None, None) == 1


if __name__ == "__main__":
    import sys
    if "--update" in sys.argv:
        sys.path.insert(0, ROOT)
        from joint_tensorf_amd import _lib
        open(os.path.join(ROOT, "include", "jt_render.abi"), "w").write("%d %s\n" % (_lib.header_version(), abi_fingerprint()))
        print(open(os.path.join(ROOT, "include", "jt_render.abi")).read())
