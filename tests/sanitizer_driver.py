"""Runs inside a python whose libjt_render.so is the AddressSanitizer / UBSan build (tests/test_sanitizers.py starts it with
LD_PRELOAD = the sanitizer runtime and JT_LIB_PATH = tests/lib/libjt_render_asan.so).  Host code only: every entry point is
called the way a careless caller would -- null pointers, zero / negative / huge sizes, unsupported scenes -- and the
workspace carves are walked at boundary capacities.  Any heap / stack / global overrun, use after free or undefined
behaviour in the library's host side aborts the process (halt_on_error); the parent checks the exit code."""
import ctypes
import itertools
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from joint_tensorf_amd import _lib  # noqa: E402

lib = _lib.lib
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def scene(kind, S=64, grid=(20, 20, 20)):
    s = _lib.JtScene()
    for a in range(3):
        s.aabb_lo[a], s.aabb_hi[a] = -1.5, 1.5
        s.plane_h[a] = s.plane_w[a] = s.line_len[a] = grid[a]
    s.n_comp_density = 16
    s.n_samples = S
    s.view_pe = s.fea_pe = 2
    if kind == "blender":
        s.n_comp_app, s.app_dim, s.mlp_hidden, s.mlp_kind = 48, 27, 64, 0
    elif kind == "llff":
        s.n_comp_app, s.app_dim, s.mlp_hidden, s.mlp_kind = 20, 20, 32, 1
    else:
        s.n_comp_app, s.app_dim, s.mlp_hidden, s.mlp_kind = 7, 5, 13, 0   # a shape no kernel is instantiated for
    return s


def zero_args(argtypes):
    out = []
    for t in argtypes:
        if t in (P, _lib.SP, _lib.FP, _lib.MP) or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            out.append(None)
        elif t is F:
            out.append(0.0)
        else:
            out.append(0)
    return out


def main():
    n_calls = 0
    # ---- every entry point on nothing at all: all pointers NULL, all sizes 0 ------------------------------------------------
    setters = {"jt_set_deterministic", "jt_shade_set_matrix_mode", "jt_shade_set_chunk_log2", "jt_shade_set_bwd_split",
               "jt_shade_set_lean_tape"}
    for name, (res, args) in sorted(_lib.SIGNATURES.items()):
        if name in setters:
            continue
        rc = getattr(lib, name)(*zero_args(args))
        n_calls += 1
        assert isinstance(rc, int), (name, rc)
    # ... and with a scene struct but nothing behind it (the argument checks that follow make_dev)
    for kind in ("blender", "llff", "other"):
        s = scene(kind)
        for name, (res, args) in sorted(_lib.SIGNATURES.items()):
            if name in setters or not args or args[0] is not _lib.SP:
                continue
            a = zero_args(args)
            a[0] = ctypes.byref(s)
            for size in (0, 1, -1, 2 ** 31 - 1):
                b = [size if (t is I and i > 0) else v for i, (t, v) in enumerate(zip(args, a))]
                rc = getattr(lib, name)(*b)
                n_calls += 1
                if res is I and name not in ("jt_shade_record_layout", "jt_shade_workspace_layout"):
                    assert rc != 0 or size <= 0 or kind == "other" or name.endswith("_bytes"), (name, kind, size, rc)
    # ---- workspace sizes and carves at boundary capacities -------------------------------------------------------------------
    out = (ctypes.c_int64 * 23)()
    prev = (lib.jt_shade_bwd_split(), lib.jt_shade_lean_tape(), lib.jt_shade_chunk_entries())
    caps = [1, 2, 31, 32, 33, 2 ** 16 - 1, 2 ** 16, 2 ** 16 + 1, 2 ** 22 - 1, 2 ** 22, 2 ** 22 + 1, 3 * 2 ** 22 + 5, 2 ** 30 - 1, 2 ** 30]
    n_layouts = 0
    for kind, split, lean, log2 in itertools.product(("blender", "llff"), (-1, 0, 1, 8, 16), (0, 1), (16, 20, 22)):
        lib.jt_shade_set_bwd_split(split)
        lib.jt_shade_set_lean_tape(lean)
        lib.jt_shade_set_chunk_log2(log2)
        s = scene(kind)
        chunk = lib.jt_shade_chunk_entries()
        assert chunk == 1 << log2
        for cap in caps:
            total = lib.jt_shade_workspace_bytes(ctypes.byref(s), cap)
            assert lib.jt_shade_workspace_layout(ctypes.byref(s), cap, out) == 0
            n_layouts += 1
            v = list(out)
            assert v[0] == total > 0, (kind, split, lean, cap, v[0], total)
            rows = v[5]
            full, lean_rows = (480, 272) if kind == "blender" else (280, 188)
            is_lean = bool(lean) and (split in (8, 16) or split == -1)   # (-1 resolves to the split form: the build's default)
            assert rows == (lean_rows if is_lean else full), (kind, split, lean, rows)
            tiles32 = (cap + 31) // 32
            assert v[1] == rows * 32 * tiles32 * 4                       # whole 32-sample tiles of `rows` rows
            nchunks = max((cap + chunk - 1) // chunk, 1)
            assert v[4] == nchunks
            pieces = [(0, v[1])] + [(v[2] + c * v[3], v[3]) for c in range(min(nchunks, 4))] + [(v[2] + (nchunks - 1) * v[3], v[3])]
            assert v[6] + v[7] <= v[3], (v[6], v[7], v[3])             # the scatter's dBasis slabs inside a chunk's slabs
            if v[8]:
                assert split == 1
                pieces += [(v[9 + 2 * i], v[10 + 2 * i]) for i in range(7)]
            else:
                assert split != 1
            for off, size in pieces:
                assert 0 <= off and size > 0 and off + size <= total, (kind, split, lean, cap, off, size, total)
            srt = sorted(set(pieces))
            for (o0, s0), (o1, s1) in zip(srt, srt[1:]):
                assert o0 + s0 <= o1, (kind, split, lean, cap, (o0, s0), (o1, s1))   # no two pieces overlap
        assert lib.jt_shade_workspace_layout(ctypes.byref(s), 0, out) == 1 and lib.jt_shade_workspace_bytes(ctypes.byref(s), 0) == 0
        for big in (2 ** 30 + 1, 2 ** 31 - 1):   # refused (the UBSan finding of round 6: the chunk count overflowed an int up there)
            assert lib.jt_shade_workspace_layout(ctypes.byref(s), big, out) == 1 and lib.jt_shade_workspace_bytes(ctypes.byref(s), big) == 0
    lib.jt_shade_set_bwd_split(prev[0])
    lib.jt_shade_set_lean_tape(prev[1])
    lib.jt_shade_set_chunk_log2(22)
    # the density backward's workspace: one formula serves the size query and the carve (jt_march.hip: march_bwd_ws_layout)
    for S, R in itertools.product((1, 8, 221, 1000, 1024), (1, 7, 2500, 62500, 2 ** 20)):
        s = scene("blender", S=S)
        b = lib.jt_march_backward_workspace_bytes(ctypes.byref(s), R)
        assert b >= R * S * 6 + R * 52 and b % 256 == 0, (S, R, b)
        n_calls += 1
    geo = (ctypes.c_int32 * 2)()
    assert lib.jt_chip_geometry(geo) == 0 and geo[0] > 0 and geo[1] > 0 and geo[0] % geo[1] == 0
    assert lib.jt_chip_geometry(None) == 1
    print("sanitizer driver: %d calls, %d workspace layouts checked, chip %d CUs / %d XCDs" % (n_calls, n_layouts, geo[0], geo[1]))


if __name__ == "__main__":
    main()
