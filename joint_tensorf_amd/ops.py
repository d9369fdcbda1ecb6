"""torch.autograd bridges over the C ABI (include/jt_render.h).

PyTorch is plumbing here: it owns device memory, the stream and the autograd tape; every number on
the hot path is produced by the HIP kernels behind `_lib.lib`.  Tensors cross the boundary as raw
device pointers.  VM factors cross CHANNEL-LAST ([H][W][C]); `factor_storage` returns that view of
a logical [1,C,H,W] parameter without a copy when the parameter already lives channel-last
(which is how joint_tensorf_amd.tensorf_repr allocates them).
"""
import ctypes
import math
import os

import torch

from . import _lib
from ._lib import JtBlurItem, JtFactors, JtMlp, JtScene, check, lib, ptr

MAT_MODE = ((0, 1), (0, 2), (1, 2))
VEC_MODE = (2, 1, 0)


def _stream():
    """raw hipStream_t of torch's current stream (the fast accessor; torch.cuda.current_stream() builds a
    Python Stream object per call, ~10 us, and the hot loop asks ~20 times per iteration)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


# JT_POSE_MARCH=0: the pose-only render keeps the plain march, its backward gathers the density taps a second time
POSE_MARCH_DERIVATIVES = os.environ.get("JT_POSE_MARCH", "1") != "0"
# JT_AUTOGRAD_THREAD=1: the engine's device thread runs the backward, as torch does by default
BACKWARD_ON_CALLER = os.environ.get("JT_AUTOGRAD_THREAD", "0") != "1"


def backward(loss, gradient=None):
    """`loss.backward(gradient)` with the autograd engine on the CALLING thread.  By default torch hands every node of a GPU graph
    to a per-device worker thread: the iteration's backward is then a handful of Python functions (this module's) run by another
    thread under the same GIL while the caller sleeps -- two thread hand-overs and their wake-up latencies per iteration, and
    nothing runs concurrently.  Measured on MI355X hosts, eager training step of the converged scene: 1.35-1.47 ms with the
    worker thread, 1.04-1.10 ms without (tools/round6/r6x3.sh).  Same nodes, same order, same streams (every node runs under the
    stream guard of its forward either way)."""
    if BACKWARD_ON_CALLER:
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward(gradient=gradient)
    else:
        loss.backward(gradient=gradient)


def lattice_indices(offsets, step, nx, ny, width):
    """[ny * nx] int64 pixel indices of the all_view_rand_grid lattice (model/nerf.py:660-667) whose two offsets are in device
    memory (`offsets`: int32, the first two elements): one launch inside a replayed graph instead of five elementwise ones."""
    assert offsets.is_cuda and offsets.dtype == torch.int32 and offsets.is_contiguous() and offsets.numel() >= 2
    out = torch.empty(ny * nx, device=offsets.device, dtype=torch.int64)
    check(lib.jt_lattice_indices(ptr(offsets), int(step), int(nx), int(ny), int(width), ptr(out), _stream()), "jt_lattice_indices")
    return out


def poke_words(dst, words, offset=0):
    """Write up to 256 32-bit words (Python ints) into the device tensor `dst` (4-byte elements) at element
    `offset`, on the current stream: the values travel as launch arguments (jt_poke) -- no staging buffer, no
    synchronisation.  This is how per-iteration host scalars reach a replayed hipGraph."""
    n = len(words)
    assert dst.element_size() == 4 and dst.is_contiguous() and offset + n <= dst.numel()
    arr = (ctypes.c_uint32 * n)(*words)
    check(lib.jt_poke(ctypes.c_void_p(dst.data_ptr() + 4 * offset), arr, n, _stream()), "jt_poke")


# ---- non-finite guard (model/tensorf.py:43-44,147-151 without the per-iteration host reads) ----------------------
FINITE_POSE, FINITE_RENDER, FINITE_LOSS, FINITE_GRAD = 1, 2, 4, 8
_STATUS = {}


def device_key(dev):
    """index of a CUDA device however it is spelt: "cuda", "cuda:0", torch.device("cuda", 0) are one device -- the key of every
    per-device cache of this package (status word, regulariser scratch, VMAdam's coefficient buffer: a cache keyed by the
    spelling is filled under "cuda" before a hipGraph capture and missed under "cuda:0" inside it)"""
    d = torch.device(dev)
    return d.index if d.index is not None else torch.cuda.current_device()


def status_word(dev):
    """the device's status word (int32[1], zero until a check finds a NaN / infinity)"""
    key = device_key(dev)
    if key not in _STATUS:
        _STATUS[key] = torch.zeros(1, device=torch.device("cuda", key), dtype=torch.int32)
        # the library reports dropped fixed-point addends / out-of-range sums of its gradient scatters here (FINITE_GRAD)
        check(lib.jt_status_bind(ptr(_STATUS[key])), "jt_status_bind")
    return _STATUS[key]


def finite_check(tensors_and_bits):
    """one launch: OR `bit` into the device's status word for every (tensor, bit) that holds a non-finite value"""
    items = [(t.detach(), b) for t, b in tensors_and_bits if t is not None and t.numel() > 0]
    if not items:
        return
    keep = [t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float() for t, _ in items]
    arr = (_lib.JtFiniteItem * len(items))()
    for k, (t, (_, b)) in enumerate(zip(keep, items)):
        arr[k].data, arr[k].n, arr[k].bit = ptr(t), t.numel(), int(b)
    check(lib.jt_finite_check(arr, len(items), ptr(status_word(keep[0].device)), _stream()), "jt_finite_check")


def read_status(dev, clear=True):
    """host read of the status word (synchronises the stream)"""
    w = status_word(dev)
    v = int(w.item())
    if v and clear:
        check(lib.jt_status_clear(_stream()), "jt_status_clear")  # the word and the library's sticky bad-addend flag
    return v


def poke_floats(dst, values, offset=0):
    n = len(values)
    assert dst.dtype == torch.float32 and dst.is_contiguous() and offset + n <= dst.numel()
    arr = (ctypes.c_float * n)(*values)
    check(lib.jt_poke(ctypes.c_void_p(dst.data_ptr() + 4 * offset), ctypes.cast(arr, ctypes.c_void_p), n, _stream()),
          "jt_poke")


# shaded-sample capacity above which the render reads the actual shaded count back instead of allocating for rays x samples
TAPE_SYNC_ENTRIES = int(float(os.environ.get("JT_TAPE_SYNC_GB", "8")) * 2 ** 30 / 1920)   # (worst-case bytes per sample: the full tape)
KEEP_INTERMEDIATES = False
# roctx ranges with the reference's record_function names (model/base.py:119-153, tensorBase.py:774) when
# opt.profiling is set (Model.train_iteration switches this on): they show up in rocprofv3 --marker-trace output.
PROFILING = False


class prof_range:
    """`with prof_range("graph.forward"):` -- a roctx range (torch.cuda.nvtx is roctx on ROCm) while PROFILING, else nothing"""

    def __init__(self, name):
        self.name, self.on = name, False

    def __enter__(self):
        if PROFILING:
            try:
                torch.cuda.nvtx.range_push(self.name)
                self.on = True
            except Exception:
                self.on = False
        return self

    def __exit__(self, *exc):
        if self.on:
            torch.cuda.nvtx.range_pop()
        return False


# In-step kernel timing (bench.py's roofline): while STEP_TIMERS is a list, every training render appends
# (kind, start event, end event, shade_offset tensor) for its k_shade_fwd<train> and k_shade_bwd launches -- HIP events on
# the launch stream, read by the caller after its own synchronisation (no sync is added here).
STEP_TIMERS = None
STEP_TIMERS_WALK = False  # also time jt_march_backward and count its listed samples (one more launch per step)
_AUX = {}
_WS = {}
_WS_EPOCH = {}
_WS_GEN = [0]  # bumped whenever a persistent workspace is (re)allocated: captured hipGraphs hold its address


def _workspace(dev, name, nbytes):
    """Persistent scratch buffer (grown on demand).  Re-used across iterations instead of a fresh
    torch.empty per call: multi-GB blocks that are also touched by the auxiliary stream make the caching
    allocator fall back to hipMalloc / hipFree, which shows up as multi-ms spikes."""
    key = (str(dev), name)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("workspace %r would grow during hipGraph capture" % (name,))
        _WS[key] = t = torch.empty(max(int(nbytes), 16), device=dev, dtype=torch.uint8)
        _WS_GEN[0] += 1
    return t


def workspace_generation():
    return _WS_GEN[0]


def _workspace_claim(dev, name):
    """A new owner writes the named workspace: returns its ticket.  Whoever wants to READ what it left there
    later compares the ticket with _workspace_owner()."""
    key = (str(dev), name)
    _WS_EPOCH[key] = _WS_EPOCH.get(key, 0) + 1
    return _WS_EPOCH[key]


def _workspace_owner(dev, name):
    return _WS_EPOCH.get((str(dev), name), 0)


# weight-gradient GEMMs on an auxiliary stream, next to the density backward.  JT_NO_AUX=1: never (clean per-kernel profiles);
# JT_NO_AUX=0: always; unset: only for the 20-channel scene, whose small GEMMs (136-256 registers, 0.14 ms) fit beside the
# persistent density walk (bat_llff_VM_MLP final grid 2.29 ms with, 2.39 without).  VM-48's GEMMs (160-368 registers) and the
# walk's 8-wave workgroups keep each other off the CUs: 3.32-3.36 ms with the auxiliary stream, 3.26 without (round 4).
_AUX_ENV = os.environ.get("JT_NO_AUX")
USE_AUX_STREAM = _AUX_ENV != "1"


def _use_aux(cfg):
    """The weight-gradient GEMMs on the auxiliary stream?  Where the appearance backward runs SPLIT (chain kernel, then the
    atomic-bound scatter) they fork behind the chain and run beside the scatter: 3.12 -> 3.07 ms on VM-48 (round 5), 0.09 ms
    on the 20-channel scene (round 4).  Beside the FUSED kernel -- VM-48 with the fp32 chain, jt_shade_set_bwd_split(0) -- they
    cost 0.06-0.10 ms (their registers and the persistent walk keep each other off the CUs) and stay on the launch stream."""
    if not USE_AUX_STREAM:
        return False
    if _AUX_ENV == "0":
        return True
    split = lib.jt_shade_bwd_split()
    if split < 0:
        return cfg.n_comp_app < 48 or bool(lib.jt_shade_matrix_mode() & 4)
    return split != 0


def _aux_stream(dev):
    """per-device auxiliary stream + fork/join events for the overlapped weight-gradient GEMMs."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _AUX:
        with torch.cuda.device(key):
            _AUX[key] = (torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event())
            # touch the events once so that their handles exist
            for ev in _AUX[key][1:]:
                ev.record()
    return _AUX[key]


# Early optimizer step of the appearance factors (round 5): see RenderRays.backward.  JT_ADAM_EARLY=0 switches it off.
ADAM_EARLY = os.environ.get("JT_ADAM_EARLY", "1") != "0"
_EARLY_GRADS = {}    # device index -> (event behind the appearance backward, addresses of the final gradient tensors, their storage)
_EARLY_EVENTS = {}


def _early_event(dev):
    key = device_key(dev)
    if key not in _EARLY_EVENTS:
        with torch.cuda.device(key):
            _EARLY_EVENTS[key] = torch.cuda.Event()
    return _EARLY_EVENTS[key]


def take_early_grads(dev):
    """(event, addresses, flat gradient storage, auxiliary stream) of the gradients the last render backward declared final
    early, once; None when there are none."""
    if not _EARLY_GRADS:
        return None
    e = _EARLY_GRADS.pop(device_key(dev), None)
    return None if e is None else e + (_aux_stream(dev)[0],)


_REG_SCRATCH = {}


def _reg_scratch(dev):
    """jt_reg_losses_forward's 640 floats of device scratch: zero when the first call sees them, left zero by every call (the
    kernel's last workgroup resets them) -- one persistent buffer per device instead of a zero fill per iteration"""
    key = device_key(dev)
    if key not in _REG_SCRATCH:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the regularisers' scratch must exist before a hipGraph capture")
        _REG_SCRATCH[key] = torch.zeros(640, device=torch.device("cuda", key), dtype=torch.float32)
    return _REG_SCRATCH[key]


# value + gradient of the regularisers in one launch at forward time (RenderRays; JT_FUSE_REG=0 turns it off)
FUSE_REG_GRADIENT = os.environ.get("JT_FUSE_REG", "1") != "0"
REG_FUSION_STATS = {"trusted": 0, "rewritten": 0}


def _reg_hint_holds(g_reg, hint):
    """Is the upstream gradient of reg3 that arrived in the backward the one the forward was told to expect?  A device
    hint (hipGraph replay: the loss-weight vector in static device memory): the same three floats by ADDRESS.  Host floats:
    LossSum.backward tags the gradient it returns for a unit upstream gradient with the weights it was built from."""
    if g_reg is None or hint is None:
        return False
    if torch.is_tensor(hint):
        return g_reg.numel() == 3 and g_reg.data_ptr() == hint.data_ptr() and g_reg.is_contiguous()
    tag = getattr(g_reg, "_jt_w", None)
    return tag is not None and tuple(float(v) for v in tag) == tuple(float(v) for v in hint)


def _zeros_flat(groups, with_flat=False, unzeroed=()):
    """Zero tensors shaped like the tensors of `groups` (a list of lists), carved out of ONE flat buffer: one fill
    launch instead of one per tensor (19 per backward).  Offsets are rounded up to 16 bytes.  with_flat: also
    return the flat buffer and the [start, end) float span of every group (contiguous: one collective each).
    unzeroed: indices of groups the caller is about to OVERWRITE completely -- they are left uninitialised (the
    regularisers' gradient is written in place of the zeros, RenderRays.backward)."""
    flat_n, plan, spans = 0, [], []
    for grp in groups:
        start = flat_n
        for t in grp:
            plan.append((flat_n, t))
            flat_n += (t.numel() + 3) // 4 * 4
        spans.append((start, flat_n))
    ref = groups[0][0]
    if unzeroed:
        flat = torch.empty(flat_n, device=ref.device, dtype=ref.dtype)
        k = 0
        while k < len(groups):  # one fill per run of consecutive groups that do need their zeros
            if k in unzeroed:
                k += 1
                continue
            j = k
            while j + 1 < len(groups) and (j + 1) not in unzeroed:
                j += 1
            flat[spans[k][0]:spans[j][1]].zero_()
            k = j + 1
    else:
        flat = torch.zeros(flat_n, device=ref.device, dtype=ref.dtype)
    it = iter(plan)
    views = [[flat[o:o + t.numel()].view(t.shape) for (o, t) in (next(it) for _ in grp)] for grp in groups]
    return (views, flat, spans) if with_flat else views


# ---- ray-sharded data parallelism --------------------------------------------------------------
# With set_data_parallel(world) the renderer's backward SUM-all-reduces its own gradients as soon as each group
# is final -- appearance factors right after the shade backward, density factors after the density walk, MLP /
# basis after the weight-gradient GEMMs -- so the collectives (RCCL, asynchronous on their own stream) overlap
# the kernels that are still running.  Callers scale the RENDER loss by 1 / world; the regularisers are the
# same on every rank and are neither scaled nor reduced.  Gradients that leave through the rays (poses) are the
# caller's to reduce (dist.allreduce_gradients).
_DP = {"world": 1, "group": None, "force": False, "no_collectives": os.environ.get("JT_DP_NO_COLLECTIVES") == "1"}


class DpReducer:
    """The gradient exchange of one backward: all gradient tensors of the backward live in ONE flat buffer, group after
    group (density planes, density lines, appearance planes, appearance lines, basis + MLP; `spans[k]` = the [start,
    end) float range of group k, ops._zeros_flat), so a run of consecutive groups is one SUM-all-reduce on a slice of
    it.  The collectives are asynchronous (RCCL's own stream); `wait()` makes the current stream wait for them -- the
    host never blocks."""

    def __init__(self, gflat, spans, group=None):
        self.gflat, self.spans, self.group, self.works = gflat, spans, group, []

    def reduce(self, first, last):
        """all-reduce groups first..last (inclusive) as one collective"""
        import torch.distributed as dist
        lo, hi = self.spans[first][0], self.spans[last][1]
        _DP.setdefault("span_elems", {})[(lo, hi)] = hi - lo  # sizes of the collectives (bench.py times them alone)
        if _DP.get("no_collectives"):  # timing aid (JT_DP_NO_COLLECTIVES=1 / bench.py's allreduce_overlap_ms): the ranks' step
            return                     # WITHOUT its gradient exchange -- the parameters of the ranks drift apart
        self.works.append(dist.all_reduce(self.gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []


def set_data_parallel(world, group=None, force=False):
    """force=True issues the collectives even for a group of one rank (exercises the RCCL path on a one-GPU box)."""
    _DP["world"], _DP["group"], _DP["force"] = int(world), group, bool(force)


def data_parallel_world():
    return _DP["world"]


def factor_storage(p):
    """logical [1,C,H,W] -> contiguous [H,W,C] tensor (no copy if p is stored channel-last).  For a Parameter the
    view is remembered on the object (three tensor constructions per call otherwise, ~40 calls per iteration on a
    host-bound path); it is dropped as soon as the parameter's storage or shape is not what it was."""
    c = p.__dict__.get("_jt_cl") if isinstance(p, torch.nn.Parameter) else None
    if c is not None and c[0] == p.data_ptr() and c[1] == p.shape:
        return c[2]
    x = p.detach()[0].permute(1, 2, 0)
    x = x if x.is_contiguous() else x.contiguous()
    if isinstance(p, torch.nn.Parameter) and x.data_ptr() == p.data_ptr():
        p.__dict__["_jt_cl"] = (p.data_ptr(), p.shape, x)
    return x


def factor_logical(x):
    """[H,W,C] storage -> logical [1,C,H,W] view."""
    return x.permute(2, 0, 1)[None]


def new_factor(C, H, W, device, dtype=torch.float32):
    """zero tensor of logical shape [1,C,H,W] stored channel-last."""
    return factor_logical(torch.zeros(H, W, C, device=device, dtype=dtype))


class RenderCfg:
    """Per-call description of the scene (mirrors JtScene)."""

    def __init__(self, aabb, plane_hw, line_len, n_comp_density, n_comp_app, step_size, near_far,
                 distance_scale, density_shift, density_act, weight_thres, n_samples, ndc, white_bg,
                 app_dim, mlp_kind, mlp_hidden, view_pe, fea_pe, view_pe_progress=1.0, fea_pe_progress=1.0,
                 alpha_mask=None, near_dev=None):
        self.aabb = [float(v) for v in aabb]  # lo xyz, hi xyz
        # a one-float device tensor holding near_far[0] (JtScene.near_plane_dev: hipGraph replay under a moving near plane)
        self.near_dev = near_dev
        self.plane_hw = [(int(h), int(w)) for h, w in plane_hw]
        self.line_len = [int(v) for v in line_len]
        self.n_comp_density = int(n_comp_density)
        self.n_comp_app = int(n_comp_app)
        self.step_size = float(step_size)
        self.near_far = (float(near_far[0]), float(near_far[1]))
        self.distance_scale = float(distance_scale)
        self.density_shift = float(density_shift)
        self.density_act = int(density_act)
        self.weight_thres = float(weight_thres)
        self.n_samples = int(n_samples)
        self.ndc = int(bool(ndc))
        self.white_bg = int(bool(white_bg))
        self.app_dim = int(app_dim)
        self.mlp_kind = int(mlp_kind)
        self.mlp_hidden = int(mlp_hidden)
        self.view_pe = int(view_pe)
        self.fea_pe = int(fea_pe)
        self.view_pe_progress = float(view_pe_progress)
        self.fea_pe_progress = float(fea_pe_progress)
        # (volume [z,y,x] float32 contiguous on the device, lo [3], inv [3]) or None; see AlphaGridMask
        self.alpha_mask = alpha_mask

    def scene(self):
        s = JtScene()
        for a in range(3):
            s.aabb_lo[a] = self.aabb[a]
            s.aabb_hi[a] = self.aabb[3 + a]
            s.plane_h[a], s.plane_w[a] = self.plane_hw[a]
            s.line_len[a] = self.line_len[a]
        s.n_comp_density = self.n_comp_density
        s.n_comp_app = self.n_comp_app
        s.step_size = self.step_size
        s.near_plane, s.far_plane = self.near_far
        s.near_plane_dev = self.near_dev.data_ptr() if self.near_dev is not None else None
        s.distance_scale = self.distance_scale
        s.density_shift = self.density_shift
        s.density_act = self.density_act
        s.weight_thres = self.weight_thres
        s.n_samples = self.n_samples
        s.ndc = self.ndc
        s.white_bg = self.white_bg
        s.app_dim = self.app_dim
        s.mlp_kind = self.mlp_kind
        s.mlp_hidden = self.mlp_hidden
        s.view_pe = self.view_pe
        s.fea_pe = self.fea_pe
        s.view_pe_progress = self.view_pe_progress
        s.fea_pe_progress = self.fea_pe_progress
        if self.alpha_mask is not None:
            vol, lo, inv = self.alpha_mask
            for a in range(3):
                s.mask_dims[a] = int(vol.shape[2 - a])
                s.mask_lo[a] = float(lo[a])
                s.mask_inv[a] = float(inv[a])
        return s


def _factors_struct(dp, dl, ap, al, alpha_volume=None):
    f = JtFactors()
    f.alpha_volume = ptr(alpha_volume) if alpha_volume is not None else None
    for i in range(3):
        f.density_plane[i] = ptr(dp[i]) if dp is not None else None
        f.density_line[i] = ptr(dl[i]) if dl is not None else None
        f.app_plane[i] = ptr(ap[i]) if ap is not None else None
        f.app_line[i] = ptr(al[i]) if al is not None else None
    return f


def _mlp_struct(basis, w1, b1, w2, b2, w3, b3):
    m = JtMlp()
    m.basis, m.w1, m.b1, m.w2, m.b2, m.w3, m.b3 = (ptr(t) for t in (basis, w1, b1, w2, b2, w3, b3))
    return m


# ----------------------------------------------------------------------------------------------
# the renderer
# ----------------------------------------------------------------------------------------------
class RenderRays(torch.autograd.Function):
    """(rays, VM factors, basis, MLP) -> rgb [R,3], depth [R], opacity [R].

    Drop-in for the tensor program of BatBase.forward (model/tensorf_repr/batBase.py:44-165)."""

    @staticmethod
    def forward(ctx, cfg, rays_o, rays_d, jitter, zvals, *params):
        dp, dl, ap, al = params[0:3], params[3:6], params[6:9], params[9:12]
        basis, w1, b1, w2, b2, w3, b3 = params[12:19]
        dev = rays_o.device
        assert dev.type == "cuda", "joint_tensorf_amd renders on the GPU only (no CPU fallback)"
        status_word(dev)  # bound to the library before any backward can want to report into it
        R, S = rays_o.shape[0], cfg.n_samples
        rays_o = rays_o.detach().contiguous().float()
        rays_d = rays_d.detach().contiguous().float()
        jitter = None if jitter is None else jitter.detach().contiguous().float().view(-1)
        zvals = None if zvals is None else zvals.detach().contiguous().float().view(-1)
        sdp = [factor_storage(p) for p in dp]
        sdl = [factor_storage(p) for p in dl]
        sap = [factor_storage(p) for p in ap]
        sal = [factor_storage(p) for p in al]
        mlp_t = [t.detach().contiguous() for t in (basis, w1, b1, w2, b2, w3, b3)]
        scene = cfg.scene()
        fac = _factors_struct(sdp, sdl, sap, sal, cfg.alpha_mask[0] if cfg.alpha_mask is not None else None)
        st = _stream()

        f32 = dict(device=dev, dtype=torch.float32)
        sigma_feat = torch.empty(R, S, **f32)
        weight = torch.empty(R, S, **f32)
        tmin = torch.empty(R, **f32)
        count = torch.empty(R, device=dev, dtype=torch.int32)
        offset = torch.empty(R + 1, device=dev, dtype=torch.int32)
        sidx = torch.empty(R, S, device=dev, dtype=torch.int16)
        opacity = torch.empty(R, **f32)
        depth = torch.empty(R, **f32)
        # only the rays want a gradient (test-time pose optimisation): the march also leaves the density feature's coordinate
        # derivatives, taken from the taps it has in registers, and the backward reads them instead of gathering again
        recording = getattr(cfg, "grad_enabled", True) and any(ctx.needs_input_grad)
        ctx.pose_only = recording and not any(ctx.needs_input_grad[5:])
        dfeat_dn = None
        if ctx.pose_only and POSE_MARCH_DERIVATIVES and not bool(lib.jt_set_deterministic(-1)):
            dfeat_dn = torch.empty(3, R, S, **f32)
            check(lib.jt_march_forward_pose(scene, fac, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals), R,
                                            ptr(sigma_feat), ptr(weight), ptr(tmin), ptr(count), ptr(offset), ptr(sidx),
                                            ptr(opacity), ptr(depth), ptr(dfeat_dn), st), "jt_march_forward_pose")
        else:
            check(lib.jt_march_forward(scene, fac, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals), R,
                                       ptr(sigma_feat), ptr(weight), ptr(tmin), ptr(count), ptr(offset), ptr(sidx),
                                       ptr(opacity), ptr(depth), st), "jt_march_forward")
        ctx.dfeat_dn = dfeat_dn
        n = cap = R * S  # worst case; kernels bound themselves by shade_offset[R] on the device
        if cap > TAPE_SYNC_ENTRIES and not torch.cuda.is_current_stream_capturing():
            # a batch whose worst-case tape (1.9 KB per sample) would run into tens of GB: ONE host read of the shaded
            # count the march just produced, and everything per shaded sample -- entry lists, colours, the record
            # workspace -- is sized by it (configs[3] on one GPU: 62 500 rays x 1 000 samples = 120 GB worst case,
            # 86 GB for the all-shaded random-init field, a few GB for a trained one).  Such an iteration takes 100 ms:
            # the synchronisation is not what it waits for.  Smaller batches (every training config) keep the
            # sync-free worst-case sizing.
            # (rounded up to whole 2^22-sample backward chunks: the sizes of consecutive iterations repeat, so the caching
            #  allocator hands the same multi-GB blocks back instead of going to hipMalloc / hipFree every iteration --
            #  measured: 814 ms per 62 500-ray step with exact sizes, 100 ms with repeating ones)
            gran = int(lib.jt_shade_chunk_entries())
            n = cap = min(R * S, max(gran, (int(offset[R].item()) + gran - 1) // gran * gran))
        cap_alloc = max(cap, 1)
        eray = torch.empty(cap_alloc, device=dev, dtype=torch.int32)
        esmp = torch.empty(cap_alloc, device=dev, dtype=torch.int32)
        vdir = torch.empty(cap_alloc, 3, **f32)
        check(lib.jt_shade_list(scene, ptr(rays_d), R, ptr(offset), ptr(sidx), ptr(eray), ptr(esmp), ptr(vdir),
                                cap, st), "jt_shade_list")
        rgb_s = torch.empty(cap_alloc, 3, **f32)
        mlp = _mlp_struct(*mlp_t)
        if recording:
            # training: the forward leaves the layer inputs of every shaded sample in the (persistent)
            # workspace; the backward consumes them instead of gathering / evaluating the chain again
            nbytes = lib.jt_shade_workspace_bytes(scene, cap)
            ws = _workspace(dev, "shade", nbytes)
            ctx.ws_ticket = _workspace_claim(dev, "shade")
            # (ctx.pose_only: the light set of records)
            ws_args = (ptr(ws), nbytes, _lib.JT_SHADE_POSE_ONLY if ctx.pose_only else 0)
        else:
            ws_args = (None, 0, 0)
        timed = STEP_TIMERS is not None and ws_args[0] is not None
        if timed:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        with prof_range("compute appearance feature + Rendering"):  # tensorBase.py:774
            check(lib.jt_shade_forward(scene, fac, mlp, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals),
                                       ptr(tmin), ptr(offset), R, ptr(eray), ptr(esmp), ptr(vdir), ptr(rgb_s),
                                       cap, *ws_args, st), "jt_shade_forward")
        if timed:
            t1.record()
            STEP_TIMERS.append(("fwd", t0, t1, offset))
        rgb = torch.empty(R, 3, **f32)
        cmask = torch.empty(R, device=dev, dtype=torch.int32)
        check(lib.jt_composite_forward(scene, R, ptr(offset), ptr(sidx), ptr(weight), ptr(rgb_s), ptr(opacity),
                                       ptr(rgb), ptr(cmask), st), "jt_composite_forward")
        ctx.cfg, ctx.n, ctx.cap = cfg, n, cap
        ctx.saved = (rays_o, rays_d, jitter, zvals, sdp, sdl, sap, sal, mlp_t, sigma_feat, weight, tmin, offset,
                     sidx, eray, esmp, vdir, rgb_s, cmask)
        ctx.param_shapes = [tuple(p.shape) for p in params]
        ctx.mark_non_differentiable(depth)
        ctx.set_materialize_grads(False)  # an output nobody used arrives as None in backward, not as a zero tensor
        # which samples were shaded (device tensors, no sync): offset [R+1] exclusive scan of the per-ray counts, sidx [R,S]
        # the shaded sample indices of a ray in ascending order (uint16 stored as int16).  Read by tests and by stats code.
        cfg.shade_lists = (offset, sidx)
        if KEEP_INTERMEDIATES:  # diagnostics only (tools/diag_*.py): per-sample density feature / weight of this call
            cfg.intermediates = dict(sigma_feat=sigma_feat, weight=weight, tmin=tmin)
        # regularisers of the same factors (cfg.reg_flags = (with_tv_density, with_tv_app)): evaluated here so that
        # the backward can ADD their gradient into the render gradient's buffer (jt_reg_losses_backward,
        # accumulate = 1) -- as a separate autograd node the two contributions to every density factor meet in an
        # add kernel that allocates a third tensor (6 adds and 31 MB x 3 of traffic per iteration at 400^3)
        reg3 = None
        ctx.reg = None
        flags = getattr(cfg, "reg_flags", None)
        if flags is not None:
            hw = []
            for i in range(3):
                H, W, _ = sdp[i].shape
                hw += [H, W, sdl[i].shape[0]]
            Cd, Ca = sdp[0].shape[2], sap[0].shape[2]
            scratch = _reg_scratch(dev)
            reg3 = torch.empty(3, **f32)
            ctx.reg = (hw, Cd, Ca, bool(flags[0]), bool(flags[1]))
            ctx.pre = None
            hint = getattr(cfg, "reg_weights", None)   # dL/d reg3 as this step's loss weights will make it, if the caller knows
            nig = ctx.needs_input_grad
            if (hint is not None and FUSE_REG_GRADIENT and all(nig[5:14]) and _DP["world"] <= 1 and not _DP["force"]
                    and not bool(lib.jt_set_deterministic(-1))):
                # value AND gradient in one launch (jt_reg_losses_fused): the gradient buffers of this node's backward are
                # created HERE, the regularisers' gradient is written into them (it covers every element of the density
                # factors and, with TV on the colours, of the appearance planes: no zero fill for those), and the backward
                # lets the render gradient's atomics land on top -- provided dL/d reg3 then IS the hint (checked there;
                # otherwise the two-launch backward overwrites what was written here)
                want_mlp = any(nig[17:24])
                skip = (0, 1, 2) if ctx.reg[4] else (0, 1)
                groups = [sdp, sdl, sap, sal] + ([mlp_t] if want_mlp else [])
                outs, gflat, spans = _zeros_flat(groups, with_flat=True, unzeroed=skip)
                gfac = _factors_struct(*outs[:4])
                if torch.is_tensor(hint):
                    w_host, w_dev = None, ptr(hint)
                else:
                    w_host, w_dev = (ctypes.c_float * 3)(*[float(v) for v in hint]), None
                check(lib.jt_reg_losses_fused(fac, (ctypes.c_int32 * 9)(*hw), Cd, Ca, int(ctx.reg[3]), int(ctx.reg[4]), w_host,
                                              w_dev, gfac, ptr(scratch), ptr(reg3), st), "jt_reg_losses_fused")
                ctx.pre = (outs, gflat, spans, hint, want_mlp)
            else:
                check(lib.jt_reg_losses_forward(fac, (ctypes.c_int32 * 9)(*hw), Cd, Ca, int(bool(flags[0])),
                                                int(bool(flags[1])), ptr(scratch), ptr(reg3), st), "jt_reg_losses_forward")
        return rgb, depth, opacity, reg3

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_opacity, g_reg=None):
        cfg = ctx.cfg
        (rays_o, rays_d, jitter, zvals, sdp, sdl, sap, sal, mlp_t, sigma_feat, weight, tmin, offset, sidx, eray,
         esmp, vdir, rgb_s, cmask) = ctx.saved
        dev = rays_o.device
        R = rays_o.shape[0]
        scene = cfg.scene()
        fac = _factors_struct(sdp, sdl, sap, sal, cfg.alpha_mask[0] if cfg.alpha_mask is not None else None)
        st = _stream()
        f32 = dict(device=dev, dtype=torch.float32)
        g_rgb = (torch.zeros(R, 3, **f32) if g_rgb is None else g_rgb.contiguous().float())
        g_op = None if g_opacity is None else g_opacity.contiguous().float()
        cap, n = ctx.cap, ctx.n
        cap_alloc = max(cap, 1)
        g_rgb_s = torch.empty(cap_alloc, 3, **f32)
        check(lib.jt_composite_backward(scene, R, ptr(offset), ptr(eray), ptr(esmp), ptr(weight), ptr(cmask),
                                        ptr(g_rgb), ptr(g_rgb_s), cap, st), "jt_composite_backward")
        # which groups of gradients autograd wants (test-time pose optimisation needs the rays' only)
        nig = ctx.needs_input_grad
        want_fac = any(nig[5:17])
        want_mlp = any(nig[17:24])
        fused_mlp_zero = want_fac and want_mlp
        det = bool(lib.jt_set_deterministic(-1))  # JT_DETERMINISTIC: factor gradients summed in 64-bit fixed point
        reg_first = False
        if det and want_fac:
            if _DP["world"] > 1 or _DP["force"]:
                raise RuntimeError("JT_DETERMINISTIC is a single-process debugging mode")
            # int64 shadow buffers with the layout of the float ones; converted after the density backward
            (gdp, gdl, gap, gal), gflat, spans = _zeros_flat([sdp, sdl, sap, sal], with_flat=True)
            gflat64 = torch.zeros(gflat.numel(), device=dev, dtype=torch.int64)
            shadow = [gflat64[(v.data_ptr() - gflat.data_ptr()) // 4:][:v.numel()] for grp in (gdp, gdl, gap, gal) for v in grp]
            gfac = _factors_struct(shadow[0:3], shadow[3:6], shadow[6:9], shadow[9:12])
            gfac_float = _factors_struct(gdp, gdl, gap, gal)
            fused_mlp_zero = False
            g_mlp_z = None
        elif want_fac and getattr(ctx, "pre", None) is not None and ctx.pre[4] == fused_mlp_zero:
            # the forward created the buffers and wrote the regularisers' gradient for the weights it was told (ctx.pre)
            outs, gflat, spans, hint, _ = ctx.pre
            # (the node must not keep a second reference to the gradient tensors it is about to return: AccumulateGrad takes a
            #  gradient over as the parameter's .grad only when nobody else holds it, and CLONES it otherwise -- seven copy
            #  launches per iteration for the basis / MLP gradients)
            ctx.pre = None
            gdp, gdl, gap, gal = outs[:4]
            g_mlp_z = outs[4] if fused_mlp_zero else None
            del outs
            gfac = _factors_struct(gdp, gdl, gap, gal)
            reg_first = True
            if not _reg_hint_holds(g_reg, hint):
                # dL/d reg3 is not what the forward assumed (or nobody used reg3): the two-launch form overwrites it
                hw, Cd, Ca, wd, wa = ctx.reg
                g3c = torch.zeros(3, **f32) if g_reg is None else g_reg.contiguous().float()
                check(lib.jt_reg_losses_backward(fac, (ctypes.c_int32 * 9)(*hw), Cd, Ca, ptr(g3c), int(wd), int(wa), gfac, 0,
                                                 ptr(torch.empty(36, **f32)), st), "jt_reg_losses_backward")
                REG_FUSION_STATS["rewritten"] += 1
            else:
                REG_FUSION_STATS["trusted"] += 1
        elif want_fac:
            # gradient buffers (channel-last storage; the kernels accumulate with atomics).  Single process with the
            # regularisers in this node: their gradient covers every element of the density factors (L1) and, with TV on
            # the colours, of the appearance planes -- it is WRITTEN first, in place of those tensors' zero fill (the
            # atomics of the render backward then land on top of it; before: fill + read-modify-write of the same bytes)
            reg_first = (ctx.reg is not None and g_reg is not None and _DP["world"] <= 1 and not _DP["force"])
            skip = ((0, 1, 2) if ctx.reg[4] else (0, 1)) if reg_first else ()
            if fused_mlp_zero:
                (gdp, gdl, gap, gal, g_mlp_z), gflat, spans = _zeros_flat([sdp, sdl, sap, sal, mlp_t], with_flat=True,
                                                                          unzeroed=skip)
            else:
                (gdp, gdl, gap, gal), gflat, spans = _zeros_flat([sdp, sdl, sap, sal], with_flat=True, unzeroed=skip)
            gfac = _factors_struct(gdp, gdl, gap, gal)
            if reg_first:
                hw, Cd, Ca, wd, wa = ctx.reg
                scratch = torch.empty(36, **f32)
                g3c = g_reg.contiguous().float()
                check(lib.jt_reg_losses_backward(fac, (ctypes.c_int32 * 9)(*hw), Cd, Ca, ptr(g3c), int(wd), int(wa), gfac, 0,
                                                 ptr(scratch), st), "jt_reg_losses_backward")
        else:
            gfac = None
        g_xyz = torch.empty(cap_alloc, 3, **f32)
        join = None
        g_mlp = [None] * 7
        if _EARLY_GRADS:
            _EARLY_GRADS.pop(device_key(dev), None)  # (an earlier backward's offer nobody took)
        dp_on = _DP["world"] > 1 or _DP["force"]
        dp = dp_on and fused_mlp_zero
        if dp_on and not dp and (want_fac or want_mlp):
            raise RuntimeError("data-parallel render backward needs the fused path with all scene gradients wanted")

        reducer = DpReducer(gflat, spans, _DP["group"]) if dp else None
        mlp = _mlp_struct(*mlp_t)
        if want_mlp:
            g_mlp = g_mlp_z if fused_mlp_zero else [torch.zeros_like(t) for t in mlp_t]
            gm = _mlp_struct(*g_mlp)
        else:
            gm = None
        nbytes = lib.jt_shade_workspace_bytes(scene, cap)
        ws = _workspace(dev, "shade", nbytes)
        if _workspace_owner(dev, "shade") != ctx.ws_ticket:
            # another render wrote the workspace since this one's forward (several forwards before one
            # backward): put this call's records back
            ctx.ws_ticket = _workspace_claim(dev, "shade")
            check(lib.jt_shade_forward(scene, fac, mlp, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals),
                                       ptr(tmin), ptr(offset), R, ptr(eray), ptr(esmp), ptr(vdir),
                                       ptr(torch.empty_like(rgb_s)), cap, ptr(ws), nbytes,
                                       _lib.JT_SHADE_POSE_ONLY if ctx.pose_only else 0, st),
                  "jt_shade_forward")
        if _use_aux(cfg) and want_mlp:
            aux, ev_fork, ev_join = _aux_stream(dev)
            # the weight-gradient GEMMs read mlp_t / ws and write g_mlp on the auxiliary stream
            if not torch.cuda.is_current_stream_capturing():  # graph-pool memory is never recycled elsewhere
                for t in list(mlp_t) + g_mlp + [offset]:
                    t.record_stream(aux)
            h_aux = (ctypes.c_void_p(aux.cuda_stream), ctypes.c_void_p(ev_fork.cuda_event),
                     ctypes.c_void_p(ev_join.cuda_event))
            join = ev_join
        else:
            h_aux = (None, None, None)
        t_bwd_end = None
        if STEP_TIMERS is not None and not ctx.pose_only:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            if h_aux[0] is not None:
                # the weight-gradient GEMMs go to the auxiliary stream: what this call leaves on the launch stream IS the
                # per-sample backward (k_shade_bwd, or chain + scatter when split) -- its end mark is recorded behind the call
                t_bwd_end = t1
                # ... and the fork event the library records BEHIND THE CHAIN (split backward: the GEMMs wait for it) is a
                # timing event of this launch's own while the step timers are on: chain and scatter are told apart
                t_mid = torch.cuda.Event(enable_timing=True)
                t_mid.record()  # creates the handle
                h_aux = (h_aux[0], ctypes.c_void_p(t_mid.cuda_event), h_aux[2])
                STEP_TIMERS.append(("bwd_chain", t0, t_mid, offset))
                STEP_TIMERS.append(("bwd_scatter", t_mid, t1, offset))
            else:
                # one stream: the library records the mark between the per-sample kernels and the GEMMs (jt_render.h)
                t1.record()  # creates the handle
                h_aux = (h_aux[0], ctypes.c_void_p(t1.cuda_event), h_aux[2])
            STEP_TIMERS.append(("bwd", t0, t1, offset))
        check(lib.jt_shade_backward(scene, fac, mlp, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals),
                                    ptr(tmin), ptr(offset), R, ptr(eray), ptr(esmp), ptr(vdir), ptr(rgb_s),
                                    ptr(g_rgb_s), gfac, gm, ptr(g_xyz), cap, ptr(ws), nbytes, 0, st, *h_aux),
              "jt_shade_backward")
        if t_bwd_end is not None:
            t_bwd_end.record()
        if dp:
            reducer.reduce(2, 3)  # appearance planes + lines are final
        elif (ADAM_EARLY and want_fac and not det and _use_aux(cfg) and (reg_first or ctx.reg is None or g_reg is None)
              and not torch.cuda.is_current_stream_capturing()):
            # the appearance factors' gradients are final HERE (their regulariser part was written before the render backward);
            # what follows on this stream -- the density backward -- is bound by the float-atomic path and leaves the memory
            # system idle: an optimizer that asks (optim.VMAdam.step -> take_early_grads) steps these tensors on the auxiliary
            # stream, beside the density walk, and the rest behind it as before
            ev = _early_event(dev)
            ev.record()
            _EARLY_GRADS[device_key(dev)] = (ev, frozenset(int(t.data_ptr()) for t in list(gap) + list(gal)), gflat)
        g_o = torch.empty(R, 3, **f32)
        g_d = torch.empty(R, 3, **f32)
        mws_bytes = lib.jt_march_backward_workspace_bytes(scene, R)
        mws = _workspace(dev, "march_bwd", mws_bytes)
        timed = STEP_TIMERS is not None and STEP_TIMERS_WALK and not getattr(ctx, "pose_only", False)
        if timed:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        if getattr(ctx, "dfeat_dn", None) is not None and gfac is None:
            check(lib.jt_march_backward_pose(scene, fac, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals), R,
                                             ptr(sigma_feat), ptr(weight), ptr(tmin), ptr(offset), ptr(sidx), ptr(rgb_s),
                                             ptr(cmask), ptr(g_rgb), ptr(g_op), ptr(g_xyz), ptr(ctx.dfeat_dn), ptr(g_o), ptr(g_d),
                                             ptr(mws), mws_bytes, st), "jt_march_backward_pose")
        else:
            check(lib.jt_march_backward(scene, fac, ptr(rays_o), ptr(rays_d), ptr(jitter), ptr(zvals), R,
                                        ptr(sigma_feat), ptr(weight), ptr(tmin), ptr(offset), ptr(sidx), ptr(rgb_s),
                                        ptr(cmask), ptr(g_rgb), ptr(g_op), ptr(g_xyz), gfac, ptr(g_o), ptr(g_d),
                                        ptr(mws), mws_bytes, st), "jt_march_backward")
        if timed:
            t1.record()
            # listed samples of this call (in-box samples with a density gradient): the per-ray counts sit behind
            # gfeat [R,S] f32 and vlist [R,S] u16 in the workspace (jt_march.hip: march_bwd_ws_layout)
            S_ = cfg.n_samples
            o = (R * S_ * 4 + 255) // 256 * 256
            o = (o + R * S_ * 2 + 255) // 256 * 256
            STEP_TIMERS.append(("march_bwd", t0, t1, mws[o:o + 4 * R].view(torch.int32).sum().view(1)))
        if dp:
            reducer.reduce(0, 1)  # density planes + lines are final
        if join is not None:
            torch.cuda.current_stream().wait_event(join)  # weight gradients done before anyone reads them
        if dp:
            reducer.reduce(4, 4)  # basis + MLP
            reducer.wait()  # stream-level: whoever consumes the gradients next runs behind the collectives
        if det and want_fac:
            # fixed point -> float (value = word / 2^48), in place of the zero-filled float buffers; a sum at or beyond
            # 2^60 (value 4 096: out of the format's safe range) leaves the FINITE_GRAD bit in the device's status word --
            # read with the other non-finite checks (read_status); a non-finite ADDEND was dropped by the kernels and
            # raised the library's sticky flag, which jt_march_backward (above) has already turned into the same bit
            gflat.copy_((gflat64.double() * (1.0 / 281474976710656.0)).float())
            bad = (gflat64.abs() >= (1 << 60)).any()
            status_word(dev).bitwise_or_(bad.to(torch.int32) * FINITE_GRAD)
            gfac = gfac_float
        if reg_first:
            pass  # written before the render backward, above
        elif ctx.reg is not None and g_reg is not None and want_fac:
            # the regularisers' gradient joins the render gradient in place (after the collectives: it is the same
            # on every rank and is not part of the exchange; after the fixed-point conversion in deterministic mode)
            hw, Cd, Ca, wd, wa = ctx.reg
            scratch = torch.empty(36, **f32)
            g3c = g_reg.contiguous().float()
            check(lib.jt_reg_losses_backward(fac, (ctypes.c_int32 * 9)(*hw), Cd, Ca, ptr(g3c), int(wd), int(wa), gfac, 1,
                                             ptr(scratch), st), "jt_reg_losses_backward")
        elif ctx.reg is not None and g_reg is not None:
            raise RuntimeError("regulariser gradient wanted without factor gradients")
        g_factors = [factor_logical(t) for t in gdp + gdl + gap + gal] if want_fac else [None] * 12
        out = [None, g_o, g_d, None, None] + g_factors + list(g_mlp)
        return tuple(out)


def render_rays(cfg, rays_o, rays_d, jitter, zvals, density_plane, density_line, app_plane, app_line, basis,
                mlp_params):
    """mlp_params = (w1, b1, w2, b2, w3, b3)."""
    # inside Function.forward grad mode is always off and needs_input_grad only mirrors requires_grad: whether a
    # backward can follow (=> the forward must leave its records) is decided here
    cfg.grad_enabled = torch.is_grad_enabled()
    out = RenderRays.apply(cfg, rays_o, rays_d, jitter, zvals, *density_plane, *density_line, *app_plane,
                           *app_line, basis, *mlp_params)
    cfg.reg3 = out[3]  # None unless cfg.reg_flags asked for the regularisers
    return out[0], out[1], out[2]


# ----------------------------------------------------------------------------------------------
# single-launch render + loss + backward to the rays (test-time pose optimisation)
# ----------------------------------------------------------------------------------------------
class RenderPoseFused(torch.autograd.Function):
    """(rays of a frozen scene, supervising pixels) -> photometric loss, with d loss / d rays produced by the SAME launch
    (csrc/jt_fused.hip: one wave per ray, forward and backward fused, no tape).  Drop-in, for mode "test-optim"
    (model/bat.py:265-292), for render_rays + the render term of compute_loss + their autograd.  Returns
    (render loss = mean squared colour error over the 3 R values, rgb [R,3], depth [R], opacity [R]); only the loss is
    differentiable, and only w.r.t. rays_o / rays_d."""

    @staticmethod
    def forward(ctx, cfg, rays_o, rays_d, zvals, image, ray_idx, rays_per_view, *params):
        dp, dl, ap, al = params[0:3], params[3:6], params[6:9], params[9:12]
        dev = rays_o.device
        assert dev.type == "cuda", "joint_tensorf_amd renders on the GPU only (no CPU fallback)"
        R = rays_o.shape[0]
        rays_o = rays_o.detach().contiguous().float()
        rays_d = rays_d.detach().contiguous().float()
        zvals = None if zvals is None else zvals.detach().contiguous().float().view(-1)
        img = image.detach().contiguous().float()
        n_views = R // int(rays_per_view)
        assert img.dim() == 4 and img.shape[0] == n_views and img.shape[1] == 3 and R == n_views * int(rays_per_view)
        idx = ray_idx.detach().contiguous().to(torch.int64)
        assert idx.numel() == int(rays_per_view)
        sd = [[factor_storage(p) for p in lst] for lst in (dp, dl, ap, al)]
        mlp_t = [t.detach().contiguous() for t in params[12:19]]
        scene = cfg.scene()
        fac = _factors_struct(*sd, cfg.alpha_mask[0] if cfg.alpha_mask is not None else None)
        mlp = _mlp_struct(*mlp_t)
        nbytes = _lib.fused_lib().jt_pose_fused_workspace_bytes(scene)
        key = (str(dev), "pose_fused")
        ws = _WS.get(key)
        if ws is None or ws.numel() < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("workspace 'pose_fused' would grow during hipGraph capture")
            # zero-filled: the head of the workspace is the kernel's loss accumulator + arrival counter
            _WS[key] = ws = torch.zeros(max(int(nbytes), 16), device=dev, dtype=torch.uint8)
            _WS_GEN[0] += 1
        f32 = dict(device=dev, dtype=torch.float32)
        buf = torch.empty(R * 12 + 1, **f32)   # rgb 3 | depth | opacity | sqerr | g_o 3 | g_d 3 | loss: one allocation
        rgb, depth, opacity, sqerr = buf[0:3 * R].view(R, 3), buf[3 * R:4 * R], buf[4 * R:5 * R], buf[5 * R:6 * R]
        g_o, g_d, loss = buf[6 * R:9 * R].view(R, 3), buf[9 * R:12 * R].view(R, 3), buf[12 * R:12 * R + 1]
        check(_lib.fused_lib().jt_pose_fused(scene, fac, mlp, ptr(rays_o), ptr(rays_d), ptr(zvals), R, ptr(img), ptr(idx),
                                int(rays_per_view), int(img.shape[2] * img.shape[3]), 1.0 / (3.0 * R), ptr(rgb), ptr(depth),
                                ptr(opacity), ptr(sqerr), ptr(loss), ptr(g_o), ptr(g_d), ptr(ws), nbytes, _stream()),
              "jt_pose_fused")
        ctx.grads = (g_o, g_d)
        ctx.mark_non_differentiable(rgb, depth, opacity)
        return loss[0], rgb, depth, opacity

    @staticmethod
    def backward(ctx, g_loss, g_rgb=None, g_depth=None, g_opacity=None):
        g_o, g_d = ctx.grads
        return (None, g_o * g_loss, g_d * g_loss, None, None, None, None) + (None,) * 19


def render_pose_fused(cfg, rays_o, rays_d, zvals, image, ray_idx, rays_per_view, density_plane, density_line, app_plane,
                      app_line, basis, mlp_params):
    return RenderPoseFused.apply(cfg, rays_o, rays_d, zvals, image, ray_idx, rays_per_view, *density_plane, *density_line,
                                 *app_plane, *app_line, basis, *mlp_params)


# ----------------------------------------------------------------------------------------------
# separable blur of a factor
# ----------------------------------------------------------------------------------------------
class BlurFactor(torch.autograd.Function):
    """Replicate-padded separable blur of a logical [1,C,H,W] factor (plane, or line with W == 1).

    Replaces BAT_VMSplit.convolute_plane / convolute_line (bateRF.py:8-39).  `reinterpret=True` reproduces
    what the reference does to a plane: it reshapes the [C, g[m1], g[m0]] memory to [C, H=g[m0], W=g[m1]]
    before blurring (bateRF.py:29 with the (H, W) of the call sites bateRF.py:68,76,110,117) and returns
    [1, C, g[m0], g[m1]].  The reshape keeps every channel's flat texel index, so in
    channel-last storage it is the SAME buffer read with rows and columns exchanged in the shape only --
    no data movement; for square planes it is the identity."""

    @staticmethod
    def forward(ctx, x, taps, reinterpret):
        xs = factor_storage(x)
        H, W, C = xs.shape
        if reinterpret and W > 1:
            H, W = W, H
            xs = xs.reshape(H, W, C)
        taps = taps.detach().contiguous().float()
        out = torch.empty_like(xs)
        tmp = torch.empty_like(xs) if (H > 1 and W > 1) else None
        check(lib.jt_blur_forward(ptr(xs), ptr(out), ptr(tmp), H, W, C, ptr(taps), taps.numel(), _stream()),
              "jt_blur_forward")
        ctx.save_for_backward(taps)
        ctx.in_hw = tuple(x.shape[2:])
        return factor_logical(out)

    @staticmethod
    def backward(ctx, g):
        (taps,) = ctx.saved_tensors
        gs = factor_storage(g)
        H, W, C = gs.shape
        gin = torch.empty_like(gs)
        tmp = torch.empty_like(gs) if (H > 1 and W > 1) else None
        check(lib.jt_blur_backward(ptr(gs), ptr(gin), ptr(tmp), H, W, C, ptr(taps), taps.numel(), _stream()),
              "jt_blur_backward")
        return factor_logical(gin.reshape(ctx.in_hw[0], ctx.in_hw[1], C)), None, None


class BlurFactors(torch.autograd.Function):
    """All factor blurs of one forward call behind ONE autograd node (12 tensors: planes re-interpreted as
    the reference does, lines plain).  Same kernels as BlurFactor; this only removes per-tensor Python /
    autograd overhead, which is what bounds the early (small-grid, blur-on) stages."""

    @staticmethod
    def _items(srcs, dsts, tmps, shapes, taps):
        arr = (JtBlurItem * len(srcs))()
        for k, (a, b, t, (H, W, C), tp) in enumerate(zip(srcs, dsts, tmps, shapes, taps)):
            arr[k].in_, arr[k].out, arr[k].tmp, arr[k].taps = ptr(a), ptr(b), ptr(t), ptr(tp)
            arr[k].H, arr[k].W, arr[k].C, arr[k].n_taps = H, W, C, tp.numel()
        return arr

    @staticmethod
    def forward(ctx, taps_density, taps_color, *factors):
        td = taps_density.detach().contiguous().float()
        tc = taps_color.detach().contiguous().float()
        srcs, outs, tmps, shapes, taps, meta = [], [], [], [], [], []
        for i, x in enumerate(factors):
            is_plane = (i % 6) < 3          # order: dP0-2, dL0-2, aP0-2, aL0-2
            xs = factor_storage(x)
            H, W, C = xs.shape
            if is_plane and W > 1:
                H, W = W, H                 # the reference's reshape quirk: same memory, axes exchanged
            srcs.append(xs)
            outs.append(torch.empty_like(xs))
            tmps.append(torch.empty_like(xs) if (H > 1 and W > 1) else None)
            shapes.append((H, W, C))
            taps.append(td if i < 6 else tc)
            meta.append((tuple(xs.shape), i < 6))
        check(lib.jt_blur_batch_forward(BlurFactors._items(srcs, outs, tmps, shapes, taps), len(srcs), _stream()),
              "jt_blur_batch_forward")
        ctx.meta = (meta, shapes)
        ctx.save_for_backward(td, tc)
        return tuple(factor_logical(o.view(sh)) for o, sh in zip(outs, shapes))

    @staticmethod
    def backward(ctx, *gs):
        td, tc = ctx.saved_tensors
        meta, shapes = ctx.meta
        srcs, gins, tmps, shp, taps, idx = [], [], [], [], [], []
        for i, (g, (in_shape, dens), (H, W, C)) in enumerate(zip(gs, meta, shapes)):
            if g is None:
                continue
            gsx = factor_storage(g)
            srcs.append(gsx)
            gins.append(torch.empty_like(gsx))
            tmps.append(torch.empty_like(gsx) if (H > 1 and W > 1) else None)
            shp.append((H, W, C))
            taps.append(td if dens else tc)
            idx.append(i)
        grads = [None] * len(gs)
        if srcs:
            check(lib.jt_blur_batch_backward(BlurFactors._items(srcs, gins, tmps, shp, taps), len(srcs), _stream()),
                  "jt_blur_batch_backward")
            for i, gin in zip(idx, gins):
                grads[i] = factor_logical(gin.view(meta[i][0]))
        return (None, None) + tuple(grads)


def blur_factors(taps_density, taps_color, density_plane, density_line, app_plane, app_line):
    out = BlurFactors.apply(taps_density, taps_color, *density_plane, *density_line, *app_plane, *app_line)
    return list(out[0:3]), list(out[3:6]), list(out[6:9]), list(out[9:12])


def blur_factor(x, taps, reinterpret=False):
    return BlurFactor.apply(x, taps, reinterpret)


# ----------------------------------------------------------------------------------------------
# regularisers
# ----------------------------------------------------------------------------------------------
class FactorReg(torch.autograd.Function):
    """(sum|x|, sum_h (dx)^2, sum_w (dx)^2) of a logical [1,C,H,W] factor in one pass; the backward adds
    the three gradients, weighted by the incoming (device) scalars, in one more pass."""

    @staticmethod
    def forward(ctx, x):
        xs = factor_storage(x)
        H, W, C = xs.shape
        out = torch.zeros(3, device=xs.device, dtype=torch.float32)
        check(lib.jt_factor_reg_forward(ptr(xs), H, W, C, ptr(out), _stream()), "jt_factor_reg_forward")
        ctx.xs = xs
        return out

    @staticmethod
    def backward(ctx, g_out):
        xs = ctx.xs
        H, W, C = xs.shape
        g = torch.empty_like(xs)
        coef = g_out.contiguous().float()
        check(lib.jt_factor_reg_backward(ptr(xs), H, W, C, ptr(coef), ptr(g), 0, _stream()),
              "jt_factor_reg_backward")
        return factor_logical(g)


def factor_reg(x):
    return FactorReg.apply(x)


class RegLosses(torch.autograd.Function):
    """(L1, TV_density, TV_color) of a scene's twelve factors in one ABI call each way."""

    @staticmethod
    def forward(ctx, with_tv_density, with_tv_app, *factors):
        dp, dl, ap, al = factors[0:3], factors[3:6], factors[6:9], factors[9:12]
        st_ = [[factor_storage(p) for p in lst] for lst in (dp, dl, ap, al)]
        dev = st_[0][0].device
        hw = []
        for i in range(3):
            H, W, _ = st_[0][i].shape
            hw += [H, W, st_[1][i].shape[0]]
        hw_arr = (ctypes.c_int32 * 9)(*hw)
        fac = _factors_struct(*st_)
        scratch = _reg_scratch(dev)
        out = torch.empty(3, device=dev, dtype=torch.float32)
        Cd, Ca = st_[0][0].shape[2], st_[2][0].shape[2]
        check(lib.jt_reg_losses_forward(fac, hw_arr, Cd, Ca, int(bool(with_tv_density)), int(bool(with_tv_app)),
                                        ptr(scratch), ptr(out), _stream()), "jt_reg_losses_forward")
        ctx.saved = (st_, hw, Cd, Ca, bool(with_tv_density), bool(with_tv_app))
        return out

    @staticmethod
    def backward(ctx, g3):
        st_, hw, Cd, Ca, wd, wa = ctx.saved
        dev = st_[0][0].device
        hw_arr = (ctypes.c_int32 * 9)(*hw)
        fac = _factors_struct(*st_)
        # every element is written (accumulate = 0): no zero fill of the 31 MB / 123 MB buffers
        gd = [torch.empty_like(t) for t in st_[0]]
        gl = [torch.empty_like(t) for t in st_[1]]
        ga = [torch.empty_like(t) for t in st_[2]] if wa else [None] * 3
        gfac = _factors_struct(gd, gl, ga if wa else st_[2], st_[3])  # unused slots just need a non-null pointer
        scratch = torch.empty(36, device=dev, dtype=torch.float32)
        g3c = g3.contiguous().float()
        check(lib.jt_reg_losses_backward(fac, hw_arr, Cd, Ca, ptr(g3c), int(wd), int(wa), gfac, 0, ptr(scratch),
                                         _stream()), "jt_reg_losses_backward")
        grads = [factor_logical(t) for t in gd] + [factor_logical(t) for t in gl] + \
                [factor_logical(t) if t is not None else None for t in ga] + [None] * 3
        return (None, None) + tuple(grads)


def reg_losses(density_plane, density_line, app_plane, app_line, with_tv_density=True, with_tv_app=True):
    return RegLosses.apply(with_tv_density, with_tv_app, *density_plane, *density_line, *app_plane, *app_line)


# Upstream gradients known to be exactly 1.0 (Model._backward_seed registers its cached ones tensors here): the backward of the
# weighted loss sum is then the weights themselves, and the launch that multiplies them by 1.0 can be skipped
UNIT_SEEDS = {}


def register_unit_seed(t):
    UNIT_SEEDS[id(t)] = t
    return t


def _finite_items(check_items):
    items = [(t.detach(), b) for t, b in (check_items or ()) if t is not None and t.numel() > 0]
    keep = [t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float() for t, _ in items]
    arr = (_lib.JtFiniteItem * max(len(items), 1))()
    for k, (t, (_, b)) in enumerate(zip(keep, items)):
        arr[k].data, arr[k].n, arr[k].bit = ptr(t), t.numel(), int(b)
    return arr, len(items), keep


class LossSum(torch.autograd.Function):
    """total = w_render * render + (w_l1, w_tv_density, w_tv_color) . reg3 (Model.summarize_loss,
    model/tensorf.py:31-47) in one launch each way; as stock ops the same sum is a multiply and an add per term
    plus a select-backward (fill + copy) per regulariser -- twenty launches of a few microseconds each.
    check_items / loss_bit: the iteration's finiteness guard (finite_check) rides in the same launch."""

    _last = None   # (weights, g_render, g_reg) of the previous unit-seed backward: reused while the weights stay the same

    @staticmethod
    def forward(ctx, render, reg3, w_render, w_l1, w_tvd, w_tvc, check_items=None, loss_bit=0):
        r = render.detach().reshape(1).float()
        q = reg3.detach().contiguous().float()
        out = torch.empty(1, device=r.device, dtype=torch.float32)
        if check_items is not None:
            arr, n, keep = _finite_items(check_items)
            w = (ctypes.c_float * 4)(w_render, w_l1, w_tvd, w_tvc)
            check(lib.jt_loss_sum_check_forward(ptr(r), ptr(q), w, None, ptr(out), arr, n, int(loss_bit),
                                                ptr(status_word(r.device)), _stream()), "jt_loss_sum_check_forward")
        else:
            check(lib.jt_loss_sum_forward(ptr(r), ptr(q), w_render, w_l1, w_tvd, w_tvc, ptr(out), _stream()),
                  "jt_loss_sum_forward")
        ctx.w = (float(w_render), float(w_l1), float(w_tvd), float(w_tvc))
        ctx.render_shape = render.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        last = LossSum._last
        unit = id(g) in UNIT_SEEDS and UNIT_SEEDS[id(g)] is g
        if unit and last is not None and last[0] == (ctx.w, str(g.device)):
            # dL/dtotal is the cached ones tensor and the weights are last iteration's: so are the products
            return last[1].reshape(ctx.render_shape), last[2], None, None, None, None, None, None
        gc = g.contiguous().float().reshape(1)
        g_render = torch.empty(1, device=gc.device, dtype=torch.float32)
        g_reg = torch.empty(3, device=gc.device, dtype=torch.float32)
        check(lib.jt_loss_sum_backward(ptr(gc), *ctx.w, ptr(g_render), ptr(g_reg), _stream()), "jt_loss_sum_backward")
        if unit:
            g_reg._jt_w = ctx.w[1:]   # "these ARE the weights": RenderRays' fused regulariser gradient checks it (_reg_hint_holds)
        if unit and not torch.cuda.is_current_stream_capturing():
            LossSum._last = ((ctx.w, str(g.device)), g_render, g_reg)
        return g_render.reshape(ctx.render_shape), g_reg, None, None, None, None, None, None


def tv_depth_value(depth, n_views, grid_h, grid_w):
    """TV of the depth lattice (model/tensorf.py:126-135) as ONE launch, value only (no gradient): what the BAT yamls, which
    weight the term 0.0, need of it."""
    d = depth.detach().contiguous().float()
    assert d.numel() == n_views * grid_h * grid_w
    out = torch.empty(1, device=d.device, dtype=torch.float32)
    check(lib.jt_tv_depth_forward(ptr(d), int(n_views), int(grid_h), int(grid_w), ptr(out), _stream()), "jt_tv_depth_forward")
    return out[0]


class LossSumDyn(torch.autograd.Function):
    """LossSum with the four weights read from device memory (`w4`, rewritten by the caller with poke_floats): the
    launch arguments of a captured hipGraph stay the same while the host schedule changes the weights."""

    @staticmethod
    def forward(ctx, render, reg3, w4, check_items=None, loss_bit=0):
        r = render.detach().reshape(1).float()
        q = reg3.detach().contiguous().float()
        out = torch.empty(1, device=r.device, dtype=torch.float32)
        if check_items is not None:
            arr, n, keep = _finite_items(check_items)
            check(lib.jt_loss_sum_check_forward(ptr(r), ptr(q), None, ptr(w4), ptr(out), arr, n, int(loss_bit),
                                                ptr(status_word(r.device)), _stream()), "jt_loss_sum_check_forward")
        else:
            check(lib.jt_loss_sum_forward_dyn(ptr(r), ptr(q), ptr(w4), ptr(out), _stream()), "jt_loss_sum_forward_dyn")
        ctx.w4 = w4
        ctx.render_shape = render.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if id(g) in UNIT_SEEDS and UNIT_SEEDS[id(g)] is g:
            # dL/dtotal is the cached ones tensor: the gradients ARE the weights, which the consumers read from device memory
            return ctx.w4[0:1].reshape(ctx.render_shape), ctx.w4[1:4], None, None, None
        gc = g.contiguous().float().reshape(1)
        g_render = torch.empty(1, device=gc.device, dtype=torch.float32)
        g_reg = torch.empty(3, device=gc.device, dtype=torch.float32)
        check(lib.jt_loss_sum_backward_dyn(ptr(gc), ptr(ctx.w4), ptr(g_render), ptr(g_reg), _stream()),
              "jt_loss_sum_backward_dyn")
        return g_render.reshape(ctx.render_shape), g_reg, None, None, None


# While this is a device tensor [4] (set by graphed.GraphedTrainStep around a capture), loss_sum reads its weights from it
LOSS_WEIGHTS_STATIC = None


def loss_sum(render, reg3, w_render, w_l1, w_tv_density, w_tv_color, check_items=None, loss_bit=0):
    """check_items = [(tensor, status bit)] (+ loss_bit for the total itself): the finiteness guard in the same launch"""
    if LOSS_WEIGHTS_STATIC is not None:
        return LossSumDyn.apply(render, reg3, LOSS_WEIGHTS_STATIC, check_items, loss_bit)
    return LossSum.apply(render, reg3, float(w_render), float(w_l1), float(w_tv_density), float(w_tv_color), check_items,
                         loss_bit)


# While this is a device tensor int64[2] (set by graphed.GraphedTrainStep around a capture), render_loss reads the
# addresses of the supervising image buffer / edge-mask buffer from it instead of baking them into the launch arguments
SUPERVISION_SLOTS_STATIC = None


class RenderLoss(torch.autograd.Function):
    """nanmean squared error between rgb [B,r,3] and the GT pixels image[:, :, ray_idx], optionally with the
    hard edge-mask split (model/tensorf.py:112-124, base.py:259-261) -- one kernel each way."""

    @staticmethod
    def forward(ctx, rgb, image, ray_idx, edge_mask, edge_factor, non_edge_factor):
        rgb_c = rgb.detach().contiguous().float()
        B, r = rgb_c.shape[0], rgb_c.shape[1]
        img = image.detach().contiguous().float().view(B, 3, -1)
        idx = ray_idx.detach().contiguous().to(torch.int64)
        m = None if edge_mask is None else edge_mask.detach().contiguous().to(torch.uint8)
        dev = rgb_c.device
        acc = torch.empty(4, device=dev, dtype=torch.float32)
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        slots = SUPERVISION_SLOTS_STATIC
        if slots is not None:
            # the caller's slots name buffers of exactly this layout (it pokes img / m -like tensors' addresses)
            assert img.data_ptr() == image.data_ptr() and (m is None or m.data_ptr() == edge_mask.data_ptr())
            check(lib.jt_render_loss_forward_ind(ptr(rgb_c), ptr(slots), ptr(idx), int(m is not None), B, r, img.shape[2],
                                                 float(edge_factor), float(non_edge_factor), ptr(acc), ptr(loss),
                                                 _stream()), "jt_render_loss_forward_ind")
        else:
            check(lib.jt_render_loss_forward(ptr(rgb_c), ptr(img), ptr(idx), ptr(m), B, r, img.shape[2],
                                             float(edge_factor), float(non_edge_factor), ptr(acc), ptr(loss), _stream()),
                  "jt_render_loss_forward")
        ctx.saved = (rgb_c, img, idx, m, acc, float(edge_factor), float(non_edge_factor), slots)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        rgb_c, img, idx, m, acc, fe, fne, slots = ctx.saved
        B, r = rgb_c.shape[0], rgb_c.shape[1]
        gc = g.contiguous().float().view(1)
        g_rgb = torch.empty_like(rgb_c)
        if slots is not None:
            check(lib.jt_render_loss_backward_ind(ptr(rgb_c), ptr(slots), ptr(idx), int(m is not None), B, r,
                                                  img.shape[2], fe, fne, ptr(acc), ptr(gc), ptr(g_rgb), _stream()),
                  "jt_render_loss_backward_ind")
        else:
            check(lib.jt_render_loss_backward(ptr(rgb_c), ptr(img), ptr(idx), ptr(m), B, r, img.shape[2], fe, fne,
                                              ptr(acc), ptr(gc), ptr(g_rgb), _stream()), "jt_render_loss_backward")
        return g_rgb, None, None, None, None, None


def render_loss(rgb, image, ray_idx, edge_mask=None, edge_factor=1.0, non_edge_factor=1.0):
    return RenderLoss.apply(rgb, image, ray_idx, edge_mask, edge_factor, non_edge_factor)


# ----------------------------------------------------------------------------------------------
# camera
# ----------------------------------------------------------------------------------------------
class TrainPose(torch.autograd.Function):
    """pose = exp(se3) o noise o gt   (model/bat.py:341-353, camera.py:81-99)."""

    @staticmethod
    def forward(ctx, se3, noise, gt):
        se3c = se3.detach().contiguous().float()
        B = se3c.shape[0]
        noise_c = None if noise is None else noise.detach().contiguous().float()
        gt_c = gt.detach().contiguous().float()
        stride = 12 if gt_c.dim() == 3 else 0
        pose = torch.empty(B, 3, 4, device=se3c.device, dtype=torch.float32)
        check(lib.jt_pose_forward(ptr(se3c), ptr(noise_c), ptr(gt_c), stride, B, ptr(pose), _stream()),
              "jt_pose_forward")
        ctx.saved = (se3c, noise_c, gt_c, stride)
        return pose

    @staticmethod
    def backward(ctx, g_pose):
        se3c, noise_c, gt_c, stride = ctx.saved
        B = se3c.shape[0]
        g = g_pose.contiguous().float()
        g_se3 = torch.empty_like(se3c)
        check(lib.jt_pose_backward(ptr(se3c), ptr(noise_c), ptr(gt_c), stride, B, ptr(g), ptr(g_se3), _stream()),
              "jt_pose_backward")
        return g_se3, None, None


def train_pose(se3, noise, gt):
    return TrainPose.apply(se3, noise, gt)


_CONTIG_MEMO = {}


def _contig_cached(t):
    """t.detach().contiguous().float() for the small per-dataset constants (intrinsics and their inverses: torch's batched
    inverse hands back column-major matrices, so `.contiguous()` on it was a copy launch in EVERY iteration), remembered per
    source tensor (identity, version, layout); the source is kept alive with its copy so that an address cannot be reused."""
    if t is None:
        return None
    if t.is_contiguous() and t.dtype == torch.float32:
        return t.detach()
    if torch.is_grad_enabled() and t.requires_grad:
        return t.detach().contiguous().float()
    key = (id(t), t.data_ptr(), t._version, tuple(t.stride()), tuple(t.shape), t.dtype)
    hit = _CONTIG_MEMO.get(key)
    if hit is None:
        if len(_CONTIG_MEMO) >= 64:
            _CONTIG_MEMO.clear()
        hit = _CONTIG_MEMO[key] = (t, t.detach().contiguous().float())
    return hit[1]


class RayGen(torch.autograd.Function):
    """rays for the sampled pixel lattice only (camera.py:231-261 + 303-340)."""

    @staticmethod
    def forward(ctx, pose, intr_inv, intr, ray_idx, image_w, ndc, ndc_near):
        pose_c = pose.detach().contiguous().float()
        B = pose_c.shape[0]
        ki = _contig_cached(intr_inv)
        k = _contig_cached(intr)
        idx = ray_idx.detach().contiguous().to(torch.int64)
        r = idx.numel()
        o = torch.empty(B, r, 3, device=pose_c.device, dtype=torch.float32)
        d = torch.empty_like(o)
        check(lib.jt_raygen_forward(ptr(pose_c), ptr(ki), ptr(k), ptr(idx), B, r, int(image_w), int(bool(ndc)),
                                    float(ndc_near), ptr(o), ptr(d), _stream()), "jt_raygen_forward")
        ctx.saved = (pose_c, ki, k, idx, int(image_w), int(bool(ndc)), float(ndc_near))
        return o, d

    @staticmethod
    def backward(ctx, g_o, g_d):
        pose_c, ki, k, idx, W, ndc, near = ctx.saved
        B, r = pose_c.shape[0], idx.numel()
        g_o = g_o.contiguous().float()
        g_d = g_d.contiguous().float()
        g_pose = torch.empty(B, 3, 4, device=pose_c.device, dtype=torch.float32)
        check(lib.jt_raygen_backward(ptr(pose_c), ptr(ki), ptr(k), ptr(idx), B, r, W, ndc, near, ptr(g_o),
                                     ptr(g_d), ptr(g_pose), _stream()), "jt_raygen_backward")
        return g_pose, None, None, None, None, None, None


def ray_gen(pose, intr_inv, intr, ray_idx, image_w, ndc=False, ndc_near=1.0):
    return RayGen.apply(pose, intr_inv, intr, ray_idx, image_w, ndc, ndc_near)


class RayGenRagged(torch.autograd.Function):
    """rays of V views, every view on ITS OWN pixel list (camera.py:231-261 + 303-340; jt_raygen_*_ragged): ray_idx [n] is the
    concatenation of the views' lists, view_offset [V + 1] int32.  Returns (o [n, 3], d [n, 3])."""

    @staticmethod
    def forward(ctx, pose, intr_inv, intr, ray_idx, view_offset, image_w, ndc, ndc_near):
        pose_c = pose.detach().contiguous().float()
        V = pose_c.shape[0]
        ki = _contig_cached(intr_inv)
        k = _contig_cached(intr)
        idx = ray_idx.detach().contiguous().to(torch.int64)
        voff = view_offset.detach().contiguous().to(torch.int32)
        n = idx.numel()
        o = torch.empty(n, 3, device=pose_c.device, dtype=torch.float32)
        d = torch.empty_like(o)
        check(lib.jt_raygen_forward_ragged(ptr(pose_c), ptr(ki), ptr(k), ptr(idx), ptr(voff), V, n, int(image_w),
                                           int(bool(ndc)), float(ndc_near), ptr(o), ptr(d), _stream()),
              "jt_raygen_forward_ragged")
        ctx.saved = (pose_c, ki, k, idx, voff, int(image_w), int(bool(ndc)), float(ndc_near))
        return o, d

    @staticmethod
    def backward(ctx, g_o, g_d):
        pose_c, ki, k, idx, voff, W, ndc, near = ctx.saved
        V, n = pose_c.shape[0], idx.numel()
        g_o = g_o.contiguous().float()
        g_d = g_d.contiguous().float()
        g_pose = torch.empty(V, 3, 4, device=pose_c.device, dtype=torch.float32)
        check(lib.jt_raygen_backward_ragged(ptr(pose_c), ptr(ki), ptr(k), ptr(idx), ptr(voff), V, n, W, ndc, near, ptr(g_o),
                                            ptr(g_d), ptr(g_pose), _stream()), "jt_raygen_backward_ragged")
        return g_pose, None, None, None, None, None, None, None


def ray_gen_ragged(pose, intr_inv, intr, ray_idx, view_offset, image_w, ndc=False, ndc_near=1.0):
    return RayGenRagged.apply(pose, intr_inv, intr, ray_idx, view_offset, image_w, ndc, ndc_near)


class RenderLossViews(torch.autograd.Function):
    """one photometric nanmean PER VIEW over a ragged batch: rgb [n, 3], image [V, 3, H, W], ray_idx [n], view_offset [V + 1]
    -> loss [V] (jt_render_loss_views_*: the single-view RenderLoss bit for bit, view by view)."""

    @staticmethod
    def forward(ctx, rgb, image, ray_idx, view_offset):
        rgb_c = rgb.detach().contiguous().float()
        V = image.shape[0]
        img = image.detach().contiguous().float().view(V, 3, -1)
        idx = ray_idx.detach().contiguous().to(torch.int64)
        voff = view_offset.detach().contiguous().to(torch.int32)
        acc = torch.empty(V, 2, device=rgb_c.device, dtype=torch.float32)
        loss = torch.empty(V, device=rgb_c.device, dtype=torch.float32)
        check(lib.jt_render_loss_views_forward(ptr(rgb_c), ptr(img), ptr(idx), ptr(voff), V, img.shape[2], ptr(acc), ptr(loss),
                                               _stream()), "jt_render_loss_views_forward")
        ctx.saved = (rgb_c, img, idx, voff, acc)
        return loss

    @staticmethod
    def backward(ctx, g):
        rgb_c, img, idx, voff, acc = ctx.saved
        gc = g.contiguous().float()
        g_rgb = torch.empty_like(rgb_c)
        check(lib.jt_render_loss_views_backward(ptr(rgb_c), ptr(img), ptr(idx), ptr(voff), img.shape[0], rgb_c.shape[0],
                                                img.shape[2], ptr(acc), ptr(gc), ptr(g_rgb), _stream()),
              "jt_render_loss_views_backward")
        return g_rgb, None, None, None


def render_loss_views(rgb, image, ray_idx, view_offset):
    return RenderLossViews.apply(rgb, image, ray_idx, view_offset)


def dense_alpha(cfg, density_plane, density_line, xyz, length):
    """alpha [n] = 1 - exp(-sigma(xyz) * length) at world points xyz [n,3] (BatBase.compute_alpha, batBase.py:27-41;
    points the scene's alpha mask drops get 0)."""
    sd = [factor_storage(p) for p in density_plane]
    sl = [factor_storage(p) for p in density_line]
    fac = _factors_struct(sd, sl, None, None, cfg.alpha_mask[0] if cfg.alpha_mask is not None else None)
    xyz = xyz.detach().contiguous().float()
    out = torch.empty(xyz.shape[0], device=xyz.device, dtype=torch.float32)
    check(lib.jt_dense_alpha(cfg.scene(), fac, ptr(xyz), xyz.shape[0], float(length), ptr(out), _stream()),
          "jt_dense_alpha")
    return out


def blur_images(images, taps):
    """Separable replicate-padded blur (along W, then H) of a batch of images [n, c, H, W] with one tap vector:
    the 2-D GT blur of nerf.Model.process_GT_images (model/nerf.py:98-110) on the factor-blur kernel.  All n*c
    image planes ride the channel axis of one channel-last [H][W][n*c] tensor (padded to a multiple of 4)."""
    n, c, H, W = images.shape
    C = (n * c + 3) // 4 * 4
    x = images.new_zeros(H, W, C)
    x[:, :, :n * c] = images.reshape(n * c, H, W).permute(1, 2, 0)
    taps = taps.detach().contiguous().float()
    out, tmp = torch.empty_like(x), torch.empty_like(x)
    check(lib.jt_blur_forward(ptr(x), ptr(out), ptr(tmp), H, W, C, ptr(taps), taps.numel(), _stream()),
          "jt_blur_forward")
    return out[:, :, :n * c].permute(2, 0, 1).reshape(n, c, H, W).contiguous()


def gaussian_taps(sigma_vox, kernel_size, device):
    """kernels.get_gaussian_kernel (kernels.py:16-22): un-normalised taps clamped at 1, K even -> K+1 taps.
    ALWAYS evaluated on the host and then uploaded (`device` only says where the result goes): the eager path and the
    taps a hipGraph replay pokes into static memory (graphed.GraphedTrainStep._poke_taps, device="cpu") are the same
    floats bit for bit."""
    s = max(float(sigma_vox), 0.0001)
    ns = torch.arange(-(kernel_size // 2), kernel_size // 2 + 1, dtype=torch.float32)
    k = 1 / (s * math.sqrt(2 * math.pi)) * torch.exp(-0.5 * (ns / s) * (ns / s))
    return torch.clamp(k, max=1.0).to(device)


# ----------------------------------------------------------------------------------------------
# kernel timing probe for bench.py's roofline line
# ----------------------------------------------------------------------------------------------
class KernelProbe:
    """Times single launches of the fused appearance kernels (forward: k_shade_fwd; backward:
    k_shade_bwd without the weight-gradient pass) with HIP events on the launch stream, on the shaded
    samples of one ray batch.  Algorithmic bytes per shaded sample: forward gather
    4*3*Ca*6 B; backward = the same bytes re-read + the same bytes added to the gradients."""


    def __init__(self, tf, rays_o, rays_d, n_samples, white_bg=True, ndc=False):
        dev = rays_o.device
        g = tf.gridSize.tolist()
        self.cfg = RenderCfg(
            aabb=tf.aabb.view(-1).tolist(), plane_hw=[(g[MAT_MODE[i][1]], g[MAT_MODE[i][0]]) for i in range(3)],
            line_len=[g[VEC_MODE[i]] for i in range(3)], n_comp_density=tf.density_n_comp[0],
            n_comp_app=tf.app_n_comp[0], step_size=float(tf.stepSize), near_far=tf.near_far,
            distance_scale=tf.distance_scale, density_shift=tf.density_shift,
            density_act=_lib.JT_ACT_SOFTPLUS if tf.fea2denseAct == "softplus" else _lib.JT_ACT_RELU,
            weight_thres=tf.rayMarch_weight_thres, n_samples=n_samples, ndc=ndc, white_bg=white_bg,
            app_dim=tf.app_dim, mlp_kind=tf.renderModule.kind, mlp_hidden=tf.featureC, view_pe=tf.view_pe,
            fea_pe=tf.fea_pe)
        self.tf = tf
        self.o = rays_o.detach().contiguous().float()
        self.d = rays_d.detach().contiguous().float()
        self.jitter = torch.rand(self.o.shape[0], device=dev)
        self.dev = dev

    def run(self, reps=10):
        cfg, tf, dev = self.cfg, self.tf, self.dev
        scene = cfg.scene()
        R, S = self.o.shape[0], cfg.n_samples
        f32 = dict(device=dev, dtype=torch.float32)
        sd = [[factor_storage(p) for p in lst] for lst in (tf.density_plane, tf.density_line, tf.app_plane, tf.app_line)]
        fac = _factors_struct(*sd)
        mlp_t = [t.detach().contiguous() for t in (tf.basis_mat.weight,) + tuple(tf.renderModule.weights())]
        mlp = _mlp_struct(*mlp_t)
        st = _stream()
        sigma_feat, weight, tmin = torch.empty(R, S, **f32), torch.empty(R, S, **f32), torch.empty(R, **f32)
        count = torch.empty(R, device=dev, dtype=torch.int32)
        offset = torch.empty(R + 1, device=dev, dtype=torch.int32)
        sidx = torch.empty(R, S, device=dev, dtype=torch.int16)
        opacity, depth = torch.empty(R, **f32), torch.empty(R, **f32)
        # NDC rays share one row of z values (sample_ray_ndc); the probe takes the un-jittered row
        zv = torch.linspace(cfg.near_far[0], cfg.near_far[1], S, **f32) if cfg.ndc else None
        check(lib.jt_march_forward(scene, fac, ptr(self.o), ptr(self.d), ptr(self.jitter), ptr(zv), R, ptr(sigma_feat),
                                   ptr(weight), ptr(tmin), ptr(count), ptr(offset), ptr(sidx), ptr(opacity),
                                   ptr(depth), st), "jt_march_forward")
        n = int(offset[R].item())
        n_in_box = int((sigma_feat != 0).sum().item())
        eray = torch.empty(max(n, 1), device=dev, dtype=torch.int32)
        esmp = torch.empty(max(n, 1), device=dev, dtype=torch.int32)
        vdir = torch.empty(max(n, 1), 3, **f32)
        rgb_s = torch.empty(max(n, 1), 3, **f32)
        check(lib.jt_shade_list(scene, ptr(self.d), R, ptr(offset), ptr(sidx), ptr(eray), ptr(esmp), ptr(vdir), n, st),
              "jt_shade_list")
        bytes_per = 4 * 3 * cfg.n_comp_app * 6

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in ev:
                a.record()
                fn()
                b.record()
            torch.cuda.synchronize()
            return sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2] * 1e-3

        nbytes = lib.jt_shade_workspace_bytes(scene, max(n, 1))
        ws = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
        t_inf = timed(lambda: check(lib.jt_shade_forward(
            scene, fac, mlp, ptr(self.o), ptr(self.d), ptr(self.jitter), ptr(zv), ptr(tmin), ptr(offset), R, ptr(eray),
            ptr(esmp), ptr(vdir), ptr(rgb_s), n, None, 0, 0, st), "jt_shade_forward"))
        # training forward: also leaves the layer-input records for the backward in the workspace
        t_fwd = timed(lambda: check(lib.jt_shade_forward(
            scene, fac, mlp, ptr(self.o), ptr(self.d), ptr(self.jitter), ptr(zv), ptr(tmin), ptr(offset), R, ptr(eray),
            ptr(esmp), ptr(vdir), ptr(rgb_s), n, ptr(ws), nbytes, 0, st), "jt_shade_forward"))
        # one backward launch = one chunk of shaded samples
        nb = min(n, int(lib.jt_shade_chunk_entries()))  # one backward launch = one chunk
        g_rgb_s = torch.rand(max(n, 1), 3, **f32)
        gfac = _factors_struct(*[[torch.zeros_like(t) for t in lst] for lst in sd])
        self._keep = gfac
        gm_t = [torch.zeros_like(t) for t in mlp_t]
        gm = _mlp_struct(*gm_t)
        g_xyz = torch.empty(max(n, 1), 3, **f32)
        t_bwd = timed(lambda: check(lib.jt_shade_backward(
            scene, fac, mlp, ptr(self.o), ptr(self.d), ptr(self.jitter), ptr(zv), ptr(tmin), ptr(offset), R, ptr(eray),
            ptr(esmp), ptr(vdir), ptr(rgb_s), ptr(g_rgb_s), gfac, gm, ptr(g_xyz), nb, ptr(ws), nbytes, 1, st, None,
            None, None), "jt_shade_backward"))
        peak = 8000.0
        bwd = nb * 2 * bytes_per / t_bwd / 1e9
        fwd = n * bytes_per / t_fwd / 1e9
        return {
            "bound": "hbm", "kernel": "k_shade_bwd (one launch = one chunk of shaded samples, weight-gradient pass excluded)",
            "achieved": bwd, "peak": peak, "unit": "GB/s", "frac": bwd / peak, "traffic": None,
            "launch_ms": t_bwd * 1e3, "samples_per_launch": nb, "bytes_per_sample": 2 * bytes_per,
            "forward": {"kernel": "k_shade_fwd<train> (gather + basis + MLP + layer-input records, one launch)",
                        "achieved": fwd, "peak": peak, "unit": "GB/s", "frac": fwd / peak, "launch_ms": t_fwd * 1e3,
                        "samples_per_launch": n, "bytes_per_sample": bytes_per},
            "forward_inference": {"kernel": "k_shade_fwd<infer> (gather + basis + MLP, one launch)",
                                  "achieved": n * bytes_per / t_inf / 1e9, "peak": peak, "unit": "GB/s",
                                  "frac": n * bytes_per / t_inf / 1e9 / peak, "launch_ms": t_inf * 1e3,
                                  "samples_per_launch": n, "bytes_per_sample": bytes_per},
            "in_box_samples": n_in_box, "shaded_samples": n,
        }
