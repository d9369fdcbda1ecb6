"""hipGraph replay of the steady-state training iteration.

The reference's loop (model/base.py:154-172 around model/nerf.py:650-679) is ~75 kernel launches per iteration
issued from Python.  Once the renderer is fused the HOST is what bounds an iteration whenever the field is sparse
or the grid small (2.1-2.4 ms of Python per step against 1.5 ms of kernels at 64^3): the whole step -- pose
composition, ray generation, march, shade, composite, losses, the complete backward and the Adam launch over all
scene tensors -- is therefore captured ONCE into a hipGraph (torch.cuda.CUDAGraph is a hipGraph on ROCm) and
replayed with one launch per iteration.

What changes from one iteration to the next never enters the graph as a launch argument:
  * the lattice offsets of `all_view_rand_grid` (NumPy draws, model/nerf.py:663) live in a device int32[2]; the pixel
    lattice is computed from them inside the graph,
  * Adam's per-tensor `lr / (1 - b1^t)` and `1 / sqrt(1 - b2^t)` (the lr decays every iteration,
    model/tensorf.py:441-447) live in a device array read by `jt_adam_step_dyn`,
both written by ONE `jt_poke` launch each in front of the replay (values travel as kernel arguments: no staging
buffer, no synchronisation).  The per-ray jitter is torch's graph-safe Philox stream.  The pose optimizer (600
floats, torch's fused Adam + ExponentialLR) stays outside the graph.

Everything else that is baked into the captured launches -- grid stage, sample count, lattice SHAPE (the number of
lattice points per axis takes two values), edge-loss parity, loss weights, the supervising image buffers, the
persistent workspaces -- forms the SIGNATURE of a graph.  An iteration whose signature has a graph is replayed; one
whose signature has been seen `min_repeats` times is captured; anything else (blur active: taps and 2-D scale
change per iteration; LLFF: white-background coin, decaying TV weights, pose-gradient accumulation; data-parallel
runs) runs through `Model.train_iteration` unchanged.  Host draws are consumed in exactly the eager order, so a run
switches between the two paths without changing its random streams.
"""
import os
import time

import numpy as np
import torch

from . import ops
from .optim import VMAdam
from .options import Opt


class _Entry:
    pass


class _StageProbe:
    """What GraphedTrainStep has measured at one grid stage (see GraphedTrainStep.__init__): state is "replay_probe" (a
    window of consecutive replays is being timed), "eager_probe" (the same number of eager iterations is being timed) or
    "decided" (`choice` holds for the next REPROBE iterations)."""

    def __init__(self, lead):
        self.state, self.choice, self.n, self.t0, self.since = "replay_probe", "replay", -int(lead), 0.0, 0
        self.t_replay = self.t_eager = None
        # host time the CALLER spent between the iterations of the window being timed: a window that contains a
        # validation render, a checkpoint or a status read is thrown away instead of deciding the next REPROBE iterations
        self.gaps, self.t_exit, self.discarded = [], None, 0

    def window_is_clean(self):
        """no gap between two iterations of the window above max(2 ms, 10 x the median gap)"""
        if not self.gaps:
            return True
        med = sorted(self.gaps)[len(self.gaps) // 2]
        return max(self.gaps) <= max(2e-3, 10.0 * med)


def has_key(o, k):
    return (k in o) and o[k] is not None


def capture(fn, pool=None):
    """Capture the launches of `fn()` into a hipGraph; returns (graph, fn's result).  A Python exception inside the
    capture is held until the capture has ended in an orderly way (ending a capture that an exception tore open
    crashes inside hipStreamEndCapture on this ROCm build) and re-raised afterwards."""
    graph = torch.cuda.CUDAGraph()
    box = {}
    kw = dict(pool=pool) if pool is not None else {}
    with torch.cuda.graph(graph, **kw):
        try:
            box["out"] = fn()
        except Exception as ex:
            box["err"] = ex
    if "err" in box:
        raise box["err"]
    return graph, box["out"]


class GraphedEvalRender:
    """The sliced full-image render of one view (Graph.render_by_slices, model/nerf.py:728-740) as ONE hipGraph:
    BASELINE.json configs[4] ("hipGraph-captured ray tiles").  Pose and intrinsics live in static device buffers, so
    the graph of a scene state is replayed for every held-out view; it is re-captured when the scene tensors, the
    grid stage, the blur / PE schedule position or a workspace move.  No gradient is involved."""

    def __init__(self, graph_module):
        self.g = graph_module
        self.key = None
        self.entry = None

    def _key(self, opt, pose):
        nerf = self.g.nerf
        tf = nerf.tensorf
        ptrs = tuple(int(p.data_ptr()) for p in nerf.parameters())
        return (tuple(pose.shape), int(opt.H), int(opt.W), tuple(nerf.resolution), int(nerf.n_samples),
                float(nerf.progress_host), ops.workspace_generation(), tuple(float(v) for v in tf.near_far), ptrs,
                tf.alphaMask is not None, int(opt.nerf.n_rays),
                int(opt.nerf.eval_slice_rays) if ("eval_slice_rays" in opt.nerf) else 0)

    @torch.no_grad()
    def render(self, opt, pose, intr_inv, intr):
        key = self._key(opt, pose)
        if key != self.key:
            self.entry = self.key = None
            e = _Entry()
            e.pose, e.intr_inv, e.intr = pose.clone(), intr_inv.clone(), intr.clone()
            # once eagerly: lazy initialisation and the persistent workspaces happen outside the capture
            self.g.render_by_slices(opt, e.pose, intr_inv=e.intr_inv, mode="eval", intr=e.intr)
            if ops.workspace_generation() != key[6]:
                key = self._key(opt, pose)
            e.graph, e.out = capture(lambda: self.g.render_by_slices(opt, e.pose, intr_inv=e.intr_inv, mode="eval",
                                                                     intr=e.intr))
            self.entry, self.key = e, key
        e = self.entry
        e.pose.copy_(pose)
        e.intr_inv.copy_(intr_inv)
        e.intr.copy_(intr)
        e.graph.replay()
        return Opt({k: v.clone() for k, v in e.out.items()})


class GraphedTrainStep:
    WINDOW = 32        # iterations per timed window
    WINDOW_LEAD = 4    # untimed iterations in front of each window (the first eager iterations after a run of replays refill the
    #                    caching allocator; the first replays follow eager iterations whose launches are still queued)
    REPROBE = 3000     # iterations a choice holds before the two windows are timed again (the workload of a stage changes as
    #                    the scene converges and the alpha mask shrinks the box)
    EAGER_GAIN = 0.97  # eager is chosen where its window takes less than this fraction of the replayed one

    def __init__(self, model, min_repeats=2, max_graphs=64, adaptive=None):
        """adaptive (default on; JT_GRAPH_ADAPTIVE=0 or adaptive=False: replay wherever a graph exists): replay wins where the
        host bounds the iteration (small grids, sparse converged scenes: 1.56 ms replayed against 2.04-2.12 eager at the
        128^3 stage), the eager launch where the GPU does (the late, large grids: the host's launches are hidden behind the
        kernels anyway, and the eager backward runs its weight-gradient GEMMs beside the factor scatter on a second stream,
        which a replayed graph -- fork captured or not -- does not do as well: over the same iterations, round 5, VM-48 final
        grid 3.14 eager / 3.18 replayed / 3.22 replayed with the fork captured, LLFF final grid 2.17 / 2.28 / 2.26 ms).
        Which regime a stage is in is MEASURED: once its graphs exist, WINDOW consecutive replays are timed between two
        synchronisations, then WINDOW eager iterations; the faster mode runs for REPROBE iterations, then the two windows are
        timed again.  Host draws are consumed identically on both paths, so the choice changes no random stream."""
        self.model = model
        self.adaptive = (os.environ.get("JT_GRAPH_ADAPTIVE", "1") != "0") if adaptive is None else bool(adaptive)
        self.policy = {}      # stage key -> _StageProbe
        self._timed = None    # the probe whose window is being timed right now (train_iteration notes the caller's gaps in it)
        self.decisions = []   # (iteration, grid, choice, ms eager, ms replayed) per timed pair of windows (tools/converge.py)
        self.min_repeats = int(min_repeats)
        self.max_graphs = int(max_graphs)
        self.cache = {}
        self.seen = {}
        self.pool = None
        self.epoch = None
        self.last_var = None
        self.eager_rays = {}  # (grid, samples per ray) -> most rays an EAGER step has rendered: the persistent
        #                       workspaces fit that many, and a capture is not allowed to grow them
        self.stats = dict(replayed=0, captured=0, eager=0, eager_by_choice=0)

    # ------------------------------------------------------------------------------------------------------
    def _eligible(self, opt):
        m = self.model
        has = lambda o, k: (k in o) and o[k] is not None  # noqa: E731
        return (opt.nerf.ray_sampling_strategy == "all_view_rand_grid"
                and opt.data.dataset in ("blender", "llff")
                and (not opt.optim.warmup_pose or m.it >= int(opt.optim.warmup_pose))   # past the pose-lr warm-up
                and (not has(opt.optim, "grad_accum_iter") or int(opt.optim.grad_accum_iter) == 1)
                and isinstance(m.optim, VMAdam)
                and ops.data_parallel_world() == 1 and not ops._DP["force"] and self.model.graph.ray_shard is None)

    def _zvals_static(self, opt, tf, S, refresh=True):
        """NDC scenes: static [S] row of un-jittered sample depths + [1] jitter scale that BAT_VMSplit.forward reads while
        a graph is captured / replayed; `refresh` writes linspace(near, far, S) for the CURRENT near plane with the very
        kernel the eager path uses (bit-identical rows) -- two small launches in front of a replay.  The near plane
        follows tensorf_near_plane_schedule over the first half of an LLFF run (model/tensorf.py:230-232); round 2
        refused to capture those 25 000 iterations."""
        buf = self.__dict__.setdefault("_zv", {})
        if S not in buf:
            buf[S] = (torch.zeros(S, device=opt.device, dtype=torch.float32), torch.zeros(1, device=opt.device, dtype=torch.float32))
        base, scale = buf[S]
        if refresh:
            near, far = float(tf.near_far[0]), float(tf.near_far[1])
            torch.linspace(near, far, S, out=base)
            scale.fill_((far - near) / S)
        return base, scale

    def _blur_scheduled(self, opt):
        """True while the factor-blur schedule is above its cut-off (model/tensorf.py:208-220) whatever the random
        scale turns out to be: decided without touching the host random stream (saving and restoring NumPy's state
        costs more than the rest of this class together)."""
        from .model.bat_hip import interp_schedule
        if not (opt.model in ("bat", "bat_hip") and opt.c2f_mode != "None"):
            return False
        p = self.model.graph.nerf.progress_host
        # the random scale of the density blur is <= 1: below the cut-off without it means below the cut-off with it
        return max(interp_schedule(p, opt.c2f_schedule_color), interp_schedule(p, opt.c2f_schedule_density)) >= 0.001

    def _signature(self, opt, var, ny, nx):
        m, g = self.model, self.model.graph
        it = m.it
        has = lambda o, k: (k in o) and o[k] is not None  # noqa: E731
        edge_on = False
        if has(opt, "edge_mask_on_render_loss") and opt.edge_mask_on_render_loss:
            edge_on = (it % 2 == 0) if (has(opt, "alternate_edge_loss") and opt.alternate_edge_loss) else True
        use_edge = bool(edge_on and it < opt.edge_mask_before_iter)
        # the VALUES of the four fused loss weights reach the graph through device memory (ops.LOSS_WEIGHTS_STATIC, poked
        # per replay: LLFF's TV weights decay every iteration); which terms exist at all is structure
        weights = tuple((k, None if opt.loss_weight[k] is None else (True if k == "L1" else float(opt.loss_weight[k]) != 0.0))
                        for k in sorted(opt.loss_weight))
        tf = g.nerf.tensorf
        # (the edge masks are rebuilt every 500 iterations, model/nerf.py:172-176: their address only counts while
        # the edge-weighted loss reads them)
        ptrs = tuple(int(var[k].data_ptr()) if (k in var and torch.is_tensor(var[k])) else 0
                     for k in ("idx", "pose", "intr", "intr_inv"))
        # the supervising image set (one of five blur scales per iteration, model/nerf.py:209-227) and the edge masks
        # reach the graph through device memory (ops.SUPERVISION_SLOTS_STATIC, poked per replay): layout, not address
        ptrs += tuple((tuple(var[k].shape), str(var[k].dtype), var[k].is_contiguous())
                      if (k in var and torch.is_tensor(var[k])) else None
                      for k in ("image",) + (("train_edge_masks",) if use_edge else ()))
        view_pe = fea_pe = 1.0
        from .model.bat_hip import interp_schedule
        if has(opt, "c2f_view_pe_schedule"):
            view_pe = interp_schedule(g.nerf.progress_host, opt.c2f_view_pe_schedule)
        if has(opt, "c2f_fea_pe_schedule"):
            fea_pe = interp_schedule(g.nerf.progress_host, opt.c2f_fea_pe_schedule)
        return (id(m.optim), getattr(m.optim, "_dyn_gen", 0), ops.workspace_generation(), tuple(g.nerf.resolution),
                int(g.nerf.n_samples), int(opt.nerf.n_rays), ny, nx, use_edge, weights, ptrs, float(view_pe),
                float(fea_pe), float(getattr(m, "render_loss_scale", 1.0)), tf.alphaMask is not None,
                # NDC scenes: the near plane reaches the graph through static memory (_zvals_static), only far is structure
                (float(tf.near_far[1]),) if bool(opt.camera.ndc) else tuple(float(v) for v in tf.near_far),
                id(tf.jitter_override))

    def _lattice_shapes(self, H, W, step):
        memo = self.__dict__.setdefault("_shape_memo", {})
        key = (H, W, step)
        if key not in memo:
            memo[key] = sorted({(len(range(oy, H, step)), len(range(ox, W, step))) for oy in range(step)
                                for ox in range(step)})
        return memo[key]

    def _drop_all(self):
        self.cache.clear()
        self.seen.clear()
        self.pool = None

    # ------------------------------------------------------------------------------------------------------
    def train_iteration(self, opt, var, force_eager=False):
        """Drop-in for Model.train_iteration (same state transitions, same host draws).  Around the real work: the time the
        caller spent since the previous iteration returned, noted while a launch-mode window is being timed."""
        pr = self._timed
        if pr is not None and pr.n > 0 and pr.t_exit is not None:
            pr.gaps.append(time.perf_counter() - pr.t_exit)
        try:
            return self._train_iteration(opt, var, force_eager)
        finally:
            pr = self._timed
            if pr is not None:
                pr.t_exit = time.perf_counter()

    def _train_iteration(self, opt, var, force_eager=False):
        m, g = self.model, self.model.graph
        if force_eager or not self._eligible(opt):
            self.stats["eager"] += 1
            return self._eager(opt, var)
        g.it = m.it
        tf = g.nerf.tensorf
        if opt.data.dataset != "blender":  # what render_rays does per call (model/tensorf.py:230-232)
            from .model.bat_hip import interp_schedule
            tf.near_far[0] = interp_schedule(g.nerf.progress_host, opt.tensorf_near_plane_schedule)
            opt.nerf.depth.range[0] = tf.near_far[0]
        batch_size = len(var.idx)
        step = g.lattice_step(opt, batch_size)
        epoch = (id(m.optim), ops.workspace_generation())
        if epoch != self.epoch:  # optimizer rebuilt (grid upsampled) or a workspace moved: every graph is stale
            self._drop_all()
            self.epoch = epoch
        probe = None
        if self.adaptive:
            sk = epoch + (tuple(g.nerf.resolution), int(g.nerf.n_samples), int(opt.nerf.n_rays))
            probe = self.policy.get(sk)
            if probe is None:
                if len(self.policy) > 64:
                    self.policy.clear()
                probe = self.policy[sk] = _StageProbe(self.WINDOW_LEAD)
            if probe.state == "decided":
                probe.since += 1
                if probe.since >= self.REPROBE:
                    probe.state, probe.n = "replay_probe", -self.WINDOW_LEAD
                elif probe.choice == "eager":   # nothing below is needed, not even the signature
                    self.stats["eager"] += 1
                    self.stats["eager_by_choice"] += 1
                    return self._eager(opt, var)
            if probe.state == "eager_probe":
                return self._eager_window(probe, opt, var)
        # NumPy's state is saved (40 us) only while this iteration could still end on the eager path, i.e. until the
        # graphs of all lattice shapes of the current signature exist
        base = self._signature(opt, var, 0, 0)
        shapes = self._lattice_shapes(int(opt.H), int(opt.W), step)
        complete = False  # (whether the blur is on is only known after the draw: always keep the state)
        if not self._blur_scheduled(opt):
            complete = all((base[:6] + s + base[8:] + (None, w_)) in self.cache for s in shapes
                           for w_ in ((True,) if bool(opt.nerf.setbg_opaque) else (True, False)))
        np_state = None if complete else np.random.get_state()
        ox, oy = np.random.randint(step), np.random.randint(step)
        blur = g.resolve_blur(opt, "train")  # consumes the blur-scale draw exactly like the eager path
        # factor blur on: same graph for every (schedule value, random scale) -- the two tap vectors are static device
        # memory, rewritten below in front of the replay
        blur_key = None if blur[2] is None else (blur[2], int(blur[3]))
        # white background: static flag, or the reference's CPU coin per training call (batBase.py:154) -- drawn here
        # from the same generator; the eager fallback is handed the SAME draw
        coin = None if bool(opt.nerf.setbg_opaque) else float(torch.rand((1,)))
        wb = True if coin is None else coin < 0.5
        nx, ny = len(range(ox, opt.W, step)), len(range(oy, opt.H, step))
        sig = base[:6] + (ny, nx) + base[8:] + (blur_key, wb)
        e = self.cache.get(sig)
        if e is None:
            if len(self.seen) > 4096:
                self.seen.clear()
            n = self.seen[sig] = self.seen.get(sig, 0) + 1
            fits = batch_size * ny * nx <= self.eager_rays.get((tuple(g.nerf.resolution), int(g.nerf.n_samples)), 0)
            if n <= self.min_repeats or not fits:
                np.random.set_state(np_state)
                self.stats["eager"] += 1
                if probe is not None:
                    probe.n = -self.WINDOW_LEAD   # a replay window is made of replays only
                return self._eager(opt, var, coin)
            if os.environ.get("JT_GRAPH_DEBUG") == "1" and self.cache:
                near = min(self.cache, key=lambda k: sum(a != b for a, b in zip(k, sig)))
                print("graphed: capture #%d at it %d, differs from the nearest graph in fields %s"
                      % (self.stats["captured"] + 1, m.it, [i for i, (a, b) in enumerate(zip(near, sig)) if a != b]),
                      flush=True)
            if probe is not None:
                probe.n = -self.WINDOW_LEAD
            e = self._capture(opt, var, sig, ny, nx, step, blur, coin)
            if (id(m.optim), ops.workspace_generation()) != self.epoch or e is None:
                # capture is not allowed to move anything; if it did, start over on the eager path
                self._drop_all()
                np.random.set_state(np_state)
                self.stats["eager"] += 1
                return self._eager(opt, var, coin)
            if len(self.cache) >= self.max_graphs:  # least recently replayed graph goes (its memory returns to the pool)
                del self.cache[min(self.cache, key=lambda k: self.cache[k].last_used)]
            self.cache[sig] = e
            self.stats["captured"] += 1
        # ---- replay ------------------------------------------------------------------------------------------
        e.last_used = self.stats["replayed"]
        ops.poke_words(e.off, [ox, oy] + self._supervision_words(var, sig[8]))
        if blur_key is not None:
            self._poke_taps(opt, blur)
        if bool(opt.camera.ndc):
            self._zvals_static(opt, tf, int(g.nerf.n_samples))   # this iteration's near plane
        ops.poke_floats(self._loss_weights(opt), list(m.fused_loss_weights(opt)))
        m.optim.prepare_step(e.stepped)
        timing = probe is not None and probe.state == "replay_probe"
        if timing and probe.n == 0:
            torch.cuda.synchronize()
            probe.t0 = time.perf_counter()
            probe.gaps, probe.t_exit, self._timed = [], None, probe
        e.graph.replay()
        if timing:
            probe.n += 1
            if probe.n == self.WINDOW:
                torch.cuda.synchronize()
                self._timed = None
                if probe.window_is_clean():
                    probe.t_replay = (time.perf_counter() - probe.t0) / self.WINDOW
                    probe.state, probe.n = "eager_probe", -self.WINDOW_LEAD
                else:   # the caller did something long inside the window: time it again
                    probe.discarded += 1
                    probe.n = -self.WINDOW_LEAD
        self.stats["replayed"] += 1
        pg = m.optim_pose.param_groups[0]
        if opt.optim.warmup_pose:  # model/bat.py:98-100,108-110 (a factor of one past the warm-up, which eligibility ensures)
            pg["lr_orig"] = pg["lr"]
            pg["lr"] *= min(1, m.it / opt.optim.warmup_pose)
        m.it += 1
        w = g.se3_refine.weight
        accum = int(opt.optim.pose_grad_accum_iter) if has_key(opt.optim, "pose_grad_accum_iter") else 1
        if accum <= 1:
            if w.grad is not None:
                # a partial sum left by the accumulation period that just ended on an iteration that is not a multiple of it
                # (compress_schedule can produce such a switch): the eager path adds into it, so does this one
                w.grad.add_(e.pose_grad)
            else:
                w.grad = e.pose_grad
            m.optim_pose.step()
            w.grad = None
        else:
            # pose gradients accumulate over `accum` iterations (model/bat.py:103-106; bat_llff_VM_MLP: 8 until iteration
            # 20 000), in the eager path's order.  e.pose_grad is graph memory: the next replay overwrites it.
            if w.grad is None:
                w.grad = e.pose_grad.clone()
            else:
                w.grad.add_(e.pose_grad)
            if m.it % accum == 0:
                m.optim_pose.step()
                w.grad = None
        if opt.optim.warmup_pose:
            pg["lr"] = pg["lr_orig"]
        if m.sched_pose is not None:
            m.sched_pose.step()
        g.nerf.set_progress(m.it / opt.max_iter)
        self.last_var = e.var
        return e.loss

    def _eager_window(self, probe, opt, var):
        """One iteration of the timed eager window; the last one chooses the stage's launch mode (see __init__)."""
        if probe.n == 0:
            torch.cuda.synchronize()
            probe.t0 = time.perf_counter()
            probe.gaps, probe.t_exit, self._timed = [], None, probe
        self.stats["eager"] += 1
        self.stats["eager_by_choice"] += 1
        loss = self._eager(opt, var)
        probe.n += 1
        if probe.n == self.WINDOW:
            torch.cuda.synchronize()
            self._timed = None
            if not probe.window_is_clean():   # (see the replay window)
                probe.discarded += 1
                probe.n = -self.WINDOW_LEAD
                return loss
            probe.t_eager = (time.perf_counter() - probe.t0) / self.WINDOW
            probe.choice = "eager" if probe.t_eager < self.EAGER_GAIN * probe.t_replay else "replay"
            probe.state, probe.n, probe.since = "decided", -self.WINDOW_LEAD, 0
            nerf = self.model.graph.nerf
            self.decisions.append((int(self.model.it), tuple(int(v) for v in nerf.resolution), probe.choice,
                                   round(probe.t_eager * 1e3, 3), round(probe.t_replay * 1e3, 3)))
            if os.environ.get("JT_GRAPH_QUIET") != "1":   # one line per decision: two runs can be compared by their logs
                print("graphed: it %d grid %s -> %s (%.3f ms eager, %.3f ms replayed; %d window(s) discarded for host work "
                      "between iterations)" % (self.decisions[-1] + (probe.discarded,)), flush=True)
        return loss

    @staticmethod
    def _supervision_words(var, use_edge):
        """addresses of the iteration's supervising image buffer and edge-mask buffer as four 32-bit words"""
        img = int(var.image.data_ptr())
        msk = int(var.train_edge_masks.data_ptr()) if (use_edge and torch.is_tensor(var.get("train_edge_masks"))) else 0
        return [img & 0xffffffff, img >> 32, msk & 0xffffffff, msk >> 32]

    def _loss_weights(self, opt):
        """static [4] device tensor of the fused loss weights (ops.LossSumDyn)"""
        if getattr(self, "_lw", None) is None:
            self._lw = torch.zeros(4, device=opt.device, dtype=torch.float32)
        return self._lw

    def _eager(self, opt, var, coin=None):
        nerf = self.model.graph.nerf
        key = (tuple(nerf.resolution), int(nerf.n_samples))  # before the step: it may end with an upsampling
        tf = nerf.tensorf
        prev = tf.coin_override
        if coin is not None:
            tf.coin_override = coin   # the white-background draw this iteration already consumed
        try:
            loss = self.model.train_iteration(opt, var)
        finally:
            tf.coin_override = prev
        self.last_var = var
        if "rgb" in var:
            self.eager_rays[key] = max(self.eager_rays.get(key, 0), var.rgb.shape[0] * var.rgb.shape[1])
        return loss

    # ------------------------------------------------------------------------------------------------------
    def _taps_buffer(self, opt, ksize):
        """static [2, K+1] device tensor: row 0 the density taps, row 1 the colour taps of the iteration"""
        buf = self.__dict__.setdefault("_taps", {})
        n = int(ksize) + (1 if int(ksize) % 2 == 0 else 0)
        if n not in buf:
            buf[n] = torch.zeros(2, n, device=opt.device, dtype=torch.float32)
        return buf[n]

    def _poke_taps(self, opt, blur):
        pd, pc, mode, ksize = blur
        tf = self.model.graph.nerf.tensorf
        td = tf.get_kernel(opt, mode, pd, ksize, device="cpu")
        tc = tf.get_kernel(opt, mode, pc, ksize, device="cpu")
        buf = self._taps_buffer(opt, ksize)
        assert td.numel() == buf.shape[1] == tc.numel() and 2 * td.numel() <= 256
        ops.poke_floats(buf.view(-1), td.tolist() + tc.tolist())
        return buf

    def _capture(self, opt, var, sig, ny, nx, step, blur=(None, None, None, None), coin=None):
        m, g = self.model, self.model.graph
        dev = opt.device
        e = _Entry()
        tf = g.nerf.tensorf
        if blur[2] is not None:
            buf = self._poke_taps(opt, blur)
            tf.taps_static = (buf[0], buf[1])
        if bool(opt.camera.ndc):
            tf.zvals_static = self._zvals_static(opt, tf, int(g.nerf.n_samples))
        coin_prev = tf.coin_override
        if coin is not None:
            tf.coin_override = coin
        ops.LOSS_WEIGHTS_STATIC = self._loss_weights(opt)
        ops.status_word(dev)  # the non-finite guard's word must exist BEFORE the capture (graph-pool memory is recycled)
        ops._reg_scratch(dev)  # ... and so must the regularisers' self-resetting scratch
        # [lattice offset x, y | address of the supervising images (2 words) | address of the edge masks (2 words)]
        e.off = torch.zeros(6, device=dev, dtype=torch.int32)
        ops.poke_words(e.off, [0, 0] + self._supervision_words(var, sig[8]))
        ops.SUPERVISION_SLOTS_STATIC = e.off[2:6].view(torch.int64)
        # allocated OUTSIDE the capture: the entry must keep them alive for as long as the graph exists
        W = int(opt.W)

        def lattice(_step):
            return ops.lattice_indices(e.off, step, nx, ny, W), ny, nx

        m.optim.zero_grad()
        # pose gradients may be mid-accumulation (pose_grad_accum_iter > 1, model/bat.py:103-106): the capture must not lose
        # what the iterations before it have summed
        pose_acc = g.se3_refine.weight.grad
        g.se3_refine.weight.grad = None
        # The captured backward must not meet the parameters' cached AccumulateGrad nodes: such a node carries the
        # stream it was created on -- the legacy stream of an earlier eager iteration whose autograd graph is still
        # alive somewhere (a loss kept for logging is enough) -- and the engine would synchronise that stream with the
        # capturing one, i.e. pull it into the capture (hipStreamEndCapture then crashes).  The capture therefore runs
        # the model on fresh leaf ALIASES of the parameters (same storage, no history) and takes their gradients
        # with autograd.grad.
        from torch.nn.utils import stateless
        named = [(n, p) for n, p in g.named_parameters() if p.requires_grad]
        subs = {n: p.detach().requires_grad_(True) for n, p in named}
        np_state = np.random.get_state()
        g.lattice_override = lattice
        # one stream inside the graph: the auxiliary stream of the eager backward (weight-gradient GEMMs next to the
        # density backward) buys 0.1 ms of a dense 4.8 ms step there, but as a fork / join inside a hipGraph it COSTS
        # 0.07-0.1 ms (measured, dense and sparse scene)
        aux_was = ops.USE_AUX_STREAM
        ops.USE_AUX_STREAM = aux_was and os.environ.get("JT_GRAPH_AUX", "0") == "1"
        # dL/dtotal = the model's cached ones tensor (created BEFORE the capture): no fill in the graph, and the weighted sum's
        # backward hands the device-resident weights on as they are (ops.LossSumDyn)
        seed = m._backward_seed(torch.empty((), device=opt.device, dtype=torch.float32))
        # the optimizer's coefficient buffer must exist BEFORE the capture as well (eager steps hand their coefficients over as
        # launch arguments and never create it): allocated inside, its zero fill would be replayed behind every poke
        m.optim._dyn_buffer(torch.device(opt.device), sum(len(gr["params"]) for gr in m.optim.param_groups))
        try:
            def body():
                with stateless._reparametrize_module(g, subs):
                    v = g.forward(opt, Opt(dict(var)), mode="train")
                    loss = g.compute_loss(opt, v, mode="train")
                    loss = m.summarize_loss(opt, v, loss)
                    grads = torch.autograd.grad(loss.all, list(subs.values()), grad_outputs=[seed], allow_unused=True)
                for (_, p), gr in zip(named, grads):
                    p.grad = gr
                m.optim.launch_step()
                return v, loss

            e.graph, (v, loss) = capture(body, pool=self.pool)
        except BaseException:
            g.se3_refine.weight.grad = pose_acc
            raise
        finally:
            ops.USE_AUX_STREAM = aux_was
            g.lattice_override = None
            tf.taps_static = None
            tf.zvals_static = None
            tf.coin_override = coin_prev
            ops.LOSS_WEIGHTS_STATIC = None
            ops.SUPERVISION_SLOTS_STATIC = None
            np.random.set_state(np_state)
        if self.pool is None:
            self.pool = e.graph.pool()
        e.var, e.loss = v, loss
        e.pose_grad = g.se3_refine.weight.grad
        e.stepped = {id(p) for grp in m.optim.param_groups for p in grp["params"] if p.grad is not None}
        m.optim.zero_grad()
        g.se3_refine.weight.grad = pose_acc
        return e


class GraphedTestOptim:
    """Test-time photometric pose optimisation of a held-out view (model/bat.py:265-292) with every iteration replayed
    from a hipGraph -- the launch-bound inner loop of the evaluation (200 views x 400 iterations in bat_blender_VM).

    One persistent se(3) vector [1,6], its Adam state and static copies of the view's image / pose / intrinsics are
    re-initialised per view, so the graphs (one per lattice shape) serve the whole test set.  The iteration inside
    the graph: se3 -> SE(3) -> pose composition -> lattice rays -> march / shade (pose-only records) / composite ->
    loss -> backward to the rays -> pose -> se3 -> Adam launch (`jt_adam_step_dyn`).  Lattice offsets and the Adam
    coefficients (ExponentialLR decays the lr every iteration) are poked in front of each replay; host draws are
    consumed in the eager order.  Iterations with the factor blur on (LLFF's test_kernel_schedule) run eagerly on
    the same state."""

    def __init__(self, model):
        self.model = model
        self.se3 = None
        self.cache = {}
        self.pool = None
        self.epoch = None
        self.eager_shapes = set()   # lattice shapes that have run eagerly in the current scene state
        self.force_eager = False    # run the same loop without any replay (tests)
        self.stats = dict(replayed=0, captured=0, eager=0)

    def _setup(self, opt, var):
        dev = opt.device
        if self.se3 is None:
            self.se3 = torch.nn.Parameter(torch.zeros(1, 6, device=dev))
            # torch.optim.Adam's defaults, as in the reference (model/bat.py:270)
            self.optim = VMAdam([dict(params=[self.se3], lr=opt.optim.lr_pose)], betas=(0.9, 0.999), eps=1e-8)
            self.static = {}
            self.eye = torch.eye(3, 4, device=dev)
        for k in ("image", "pose", "intr", "intr_inv"):
            if k in var and torch.is_tensor(var[k]):
                if k not in self.static or self.static[k].shape != var[k].shape:
                    self.static[k] = var[k].clone()
                    self.cache.clear()
                else:
                    self.static[k].copy_(var[k])
        with torch.no_grad():
            self.se3.zero_()
            # the view's pose in the optimised cameras' frame, once per view, into a static buffer the graphs read
            pa = self.model.graph.aligned_pose(self.static["pose"]).contiguous()
            if "pose_aligned" not in self.static or self.static["pose_aligned"].shape != pa.shape:
                self.static["pose_aligned"] = pa.clone()
                self.cache.clear()
            else:
                self.static["pose_aligned"].copy_(pa)
        self.se3.grad = None
        st = self.optim.state[self.se3]
        if st:
            st["step"] = 0.0
            st["exp_avg"].zero_()
            st["exp_avg_sq"].zero_()
        self.optim.param_groups[0]["lr"] = float(opt.optim.lr_pose)

    def _iteration(self, opt, v, se3_leaf):
        m, g = self.model, self.model.graph
        v.pose_refine_test = ops.train_pose(se3_leaf, None, self.eye)  # se3_to_SE3 (camera.py:81-99): what the eval render reads
        v.se3_refine_test = se3_leaf    # (Graph.get_pose composes exp(se3) with the view's aligned pose from THIS tensor)
        v = g.forward(opt, v, mode="test-optim")
        loss = g.compute_loss(opt, v, mode="test-optim")
        loss = m.summarize_loss(opt, v, loss)
        return v, loss

    @torch.enable_grad()
    def run(self, opt, var):
        m, g = self.model, self.model.graph
        self._setup(opt, var)
        var.se3_refine_test = self.se3
        gamma = (opt.optim.lr_pose_test_end / opt.optim.lr_pose_test) ** (1.0 / opt.optim.test_iter)
        frozen = [p for p in g.parameters() if p.requires_grad]
        for p in frozen:
            p.requires_grad_(False)
        svar = Opt(dict(var))
        svar.update(self.static)
        batch_size = len(var.idx)
        last = None
        try:
            for it in range(opt.optim.test_iter):
                g.nerf.set_test_time_progress(it / opt.optim.test_iter)
                graphable = (opt.nerf.ray_sampling_strategy == "all_view_rand_grid" and not opt.camera.ndc
                             and bool(opt.nerf.setbg_opaque) and not self.force_eager)
                np_state = np.random.get_state() if graphable else None
                e = None
                if graphable:
                    step = g.lattice_step(opt, batch_size)
                    ox, oy = np.random.randint(step), np.random.randint(step)
                    blur = g.resolve_blur(opt, "test-optim")
                    if blur[2] is None:
                        nx, ny = len(range(ox, opt.W, step)), len(range(oy, opt.H, step))
                        e = self._graph_for(opt, svar, ny, nx, step)
                    if e is None:
                        np.random.set_state(np_state)
                if e is None:   # eager iteration on the same state
                    self.se3.grad = None
                    v, loss = self._iteration(opt, Opt(dict(svar)), self.se3)
                    ops.backward(loss.all)
                    self.optim.step()
                    last = v
                    self.stats["eager"] += 1
                else:
                    ops.poke_words(e.off, [ox, oy])
                    self.optim.prepare_step({id(self.se3)})
                    e.graph.replay()
                    last = e.var
                    self.stats["replayed"] += 1
                self.optim.param_groups[0]["lr"] *= gamma   # ExponentialLR.step()
        finally:
            for p in frozen:
                p.requires_grad_(True)
        self.se3.grad = None
        # NOTE (reproduced): the pose refinement that the eval render sees is the one from the top of the last
        # iteration, i.e. from before the last Adam step (model/bat.py:284, model/nerf.py:539)
        if last is not None:
            for k in ("pose_refine_test", "rgb", "depth", "opacity", "ray_idx", "current_pose"):
                if k in last:
                    var[k] = last[k].detach().clone()
        return var

    def _graph_for(self, opt, svar, ny, nx, step):
        m, g = self.model, self.model.graph
        nerf = g.nerf
        epoch = (ops.workspace_generation(), getattr(self.optim, "_dyn_gen", 0), tuple(nerf.resolution),
                 int(nerf.n_samples), float(nerf.progress_host), tuple(int(p.data_ptr()) for p in nerf.parameters()),
                 tuple(float(v) for v in nerf.tensorf.near_far), int(opt.nerf.n_rays))
        if epoch != self.epoch:
            self.cache.clear()
            self.eager_shapes.clear()
            self.pool = None
            self.epoch = epoch
        e = self.cache.get((ny, nx))
        if e is not None:
            return e
        if not self.optim.state[self.se3] or (ny, nx) not in self.eager_shapes:
            # first meeting of this lattice shape in this scene state: one eager iteration sizes the workspaces
            self.eager_shapes.add((ny, nx))
            return None
        dev = opt.device
        e = _Entry()
        e.off = torch.zeros(2, device=dev, dtype=torch.int32)
        W = int(opt.W)

        def lattice(_step):
            return ops.lattice_indices(e.off, step, nx, ny, W), ny, nx

        np_state = np.random.get_state()
        ws_gen = ops.workspace_generation()
        g.lattice_override = lattice
        leaf = self.se3.detach().requires_grad_(True)   # fresh leaf alias: see GraphedTrainStep._capture
        self.se3.grad = None
        # the regularisers of the FROZEN scene are the same in every iteration (the reference evaluates them per iteration and
        # adds the constants to loss.all, model/bat.py:277-279): evaluated once here, outside the graph, so that the captured
        # iteration finds them in the scene's cache (keyed by the parameters' versions) instead of carrying their launch; the
        # entry keeps the tensors alive as long as its graph reads them
        with torch.enable_grad():
            tf = g.nerf.tensorf
            tf.reg_with_tv = (float(opt.loss_weight.TV_density or 0) != 0.0, float(opt.loss_weight.TV_color or 0) != 0.0)
            e.keep_reg = tf._reg()
        m._backward_seed(torch.empty((), device=dev, dtype=torch.float32))   # the cached unit seed exists before the capture
        try:
            def body():
                v, loss = self._iteration(opt, Opt(dict(svar)), leaf)
                (gr,) = torch.autograd.grad(loss.all, [leaf], grad_outputs=[m._backward_seed(loss.all)])  # (no fill launch)
                self.se3.grad = gr
                self.optim.launch_step()
                return v, loss

            e.graph, (e.var, e.loss) = capture(body, pool=self.pool)
        finally:
            g.lattice_override = None
            np.random.set_state(np_state)
        self.se3.grad = None
        if ops.workspace_generation() != ws_gen:
            return None
        if self.pool is None:
            self.pool = e.graph.pool()
        self.cache[(ny, nx)] = e
        self.stats["captured"] += 1
        return e
