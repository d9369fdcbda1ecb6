"""Synthetic stand-in for the Blender / LLFF training data (no dataset can be downloaded here).

Produces the `var` dict the reference's data loaders hand to Graph.forward
(data/blender.py:19-91, data/llff.py:43-97): idx, image [N,3,H,W], intr, intr_inv, pose [N,3,4]
(world->camera, camera looking along +z), plus Bernoulli edge masks (model/nerf.py:115-149 output shape).
"""
import math

import numpy as np
import torch

from .options import Opt


def look_at(eye):
    eye = np.asarray(eye, dtype=np.float64)
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 0.0, 1.0])
    if abs(fwd @ up) > 0.99:
        up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], 0)
    return np.concatenate([R, (-R @ eye)[:, None]], 1).astype(np.float32)


def make_views(opt, n_views, seed=0, device="cpu", with_images=True):
    rng = np.random.RandomState(seed)
    H, W = opt.H, opt.W
    poses = []
    if opt.data.dataset == "blender":
        # cameras on the upper hemisphere of radius 4 (inside the yaml depth range [2,6]) looking at the origin
        for i in range(n_views):
            th = 2 * math.pi * rng.rand()
            ph = math.radians(5 + 75 * rng.rand())
            eye = 4.0 * np.array([math.cos(th) * math.cos(ph), math.sin(th) * math.cos(ph), math.sin(ph)])
            poses.append(look_at(eye))
        f = 0.5 * W / math.tan(0.5 * 0.69)  # camera_angle_x ~ 0.69 rad (data/blender.py:29)
    else:
        # forward-facing capture (data/llff.py:43-97 after its recentering): cameras near the origin looking along +z.
        # opt.data.llff_baseline (default 0.3) = width of the square they are spread over; with opt.data.llff_focus (a depth)
        # set, every camera is turned towards the point (0, 0, focus) as a hand-held capture is -- otherwise no rotation
        base = float(opt.data.get("llff_baseline", None) or 0.3)
        focus = opt.data.get("llff_focus", None)
        zs = float(opt.data.get("llff_zspread", None) or (1.0 / 6.0))   # depth of the camera cloud as a fraction of its width
        for i in range(n_views):
            eye = np.array([base * (rng.rand() - 0.5), base * (rng.rand() - 0.5), base * zs * (rng.rand() - 0.5)])
            R = np.eye(3)
            if focus:
                fwd = np.array([0.0, 0.0, float(focus)]) - eye
                fwd /= np.linalg.norm(fwd)
                right = np.cross(np.array([0.0, 1.0, 0.0]), fwd)
                right /= np.linalg.norm(right)
                R = np.stack([right, np.cross(fwd, right), fwd], 0)
            poses.append(np.concatenate([R, (-R @ eye)[:, None]], 1).astype(np.float32))
        f = 0.8 * W
    pose = torch.tensor(np.stack(poses))
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=torch.float32)[None].repeat(n_views, 1, 1)
    var = Opt(idx=torch.arange(n_views, device=device), pose=pose.to(device), intr=intr.to(device),
              intr_inv=intr.inverse().contiguous().to(device))
    if with_images:
        g = torch.Generator().manual_seed(seed + 1)
        var.image = torch.rand(n_views, 3, H, W, generator=g).to(device)
        var.train_edge_masks = (torch.rand(n_views, H * W, generator=g) < 0.5).to(torch.uint8).to(device)
    return var


@torch.no_grad()
def bake_blobs(tensorf, n_blobs=12, seed=0, amplitude=60.0, radius=(0.12, 0.3), background=-12.0, z_range=(0.25, 0.75)):
    """Overwrite the DENSITY factors with a few opaque Gaussian blobs (the "structured" scene).

    exp(-|p - c|^2 / 2 s^2) factorises into an (x, y) plane times a z line, i.e. one blob is exactly one rank-1
    VM component of plane 0 / line 0; the last component is the constant `background` (what a trained field's empty
    space looks like: softplus(-12 - 10) gives weights of 1e-11, far below rayMarch_weight_thres, where the
    zero feature of an untouched grid still gives 4e-6); all other density components are zeroed.  Only the few
    samples at a blob's surface are shaded -- the regime of a trained scene (a few per cent of the in-box samples), against
    the random-init field in which every in-box sample is shaded."""
    rng = np.random.RandomState(seed)
    dev = tensorf.density_plane[0].device
    lo, hi = tensorf.aabb[0].to(dev).float(), tensorf.aabb[1].to(dev).float()
    C = tensorf.density_plane[0].shape[1]
    n_blobs = min(int(n_blobs), C - 1)
    for p in list(tensorf.density_plane) + list(tensorf.density_line):
        p.zero_()
    plane, line = tensorf.density_plane[0], tensorf.density_line[0]   # plane 0: (x -> W, y -> H); line 0: z
    H, W, L = plane.shape[2], plane.shape[3], line.shape[2]
    X = lo[0] + (hi[0] - lo[0]) * torch.linspace(0, 1, W, device=dev)
    Y = lo[1] + (hi[1] - lo[1]) * torch.linspace(0, 1, H, device=dev)
    Z = lo[2] + (hi[2] - lo[2]) * torch.linspace(0, 1, L, device=dev)
    for k in range(n_blobs):
        frac = 0.25 + 0.5 * rng.rand(3)
        frac[2] = z_range[0] + (z_range[1] - z_range[0]) * (frac[2] - 0.25) / 0.5   # blob centres along z (fractions of the box)
        c = (lo + (hi - lo) * torch.tensor(frac, device=dev, dtype=torch.float32))
        s = float(radius[0] + (radius[1] - radius[0]) * rng.rand())
        plane[0, k] = amplitude * torch.exp(-((X[None, :] - c[0]) ** 2 + (Y[:, None] - c[1]) ** 2) / (2 * s * s))
        line[0, k, :, 0] = torch.exp(-((Z - c[2]) ** 2) / (2 * s * s))
    plane[0, C - 1] = background
    line[0, C - 1] = 1.0
    return n_blobs


# ---- a self-consistent scene: supervising images RENDERED from a known field at the known poses ---------------------
# (what the reference gets from its image sets: model/nerf.py:35-40 loads pictures of ONE scene taken from the poses the
#  run is scored against, model/bat.py:229-263.  Uniform-noise images, make_views' default, have no such scene behind them:
#  the joint optimisation then has nothing to recover and the field turns into fog.)
@torch.no_grad()
def bake_appearance(tensorf, seed=0, amplitude=0.7, contrast=9.0):
    """Overwrite the APPEARANCE factors with smooth low-frequency patterns (every component one plane wave over its plane
    times one cosine along its line, 0.4 - 1.6 periods across the box) and raise the contrast of the random-init MLP's
    last layer, so that the colour decoded at a point varies smoothly with position (and weakly with view direction)
    over most of [0, 1]^3 -- textured, registrable content for the blobs of bake_blobs."""
    rng = np.random.RandomState(seed + 77)
    dev = tensorf.app_plane[0].device
    for i in range(3):
        plane, line = tensorf.app_plane[i], tensorf.app_line[i]
        C, H, W, L = plane.shape[1], plane.shape[2], plane.shape[3], line.shape[2]
        u = torch.linspace(0, 1, W, device=dev)[None, :]
        v = torch.linspace(0, 1, H, device=dev)[:, None]
        w = torch.linspace(0, 1, L, device=dev)
        for c in range(C):
            fu, fv, fw = (2 * math.pi * (0.4 + 1.2 * rng.rand()) * (1 if rng.rand() < 0.5 else -1) for _ in range(3))
            pu, pw = 2 * math.pi * rng.rand(), 2 * math.pi * rng.rand()
            plane[0, c] = amplitude * torch.cos(fu * u + fv * v + pu)
            line[0, c, :, 0] = torch.cos(fw * w + pw)
    last = tensorf.renderModule.weights()[4]
    last.mul_(contrast)


@torch.no_grad()
def bake_wall(tensorf, axis_frac=0.88, thickness=0.05, amplitude=40.0, component=None):
    """An opaque slab across the box at `axis_frac` of its z extent (rank-1: constant plane 0 times a bump on line 0): closes
    a forward-facing (LLFF-like) scene, so that every ray ends on content as in a real photograph."""
    plane, line = tensorf.density_plane[0], tensorf.density_line[0]
    C, L = plane.shape[1], line.shape[2]
    c = C - 2 if component is None else component
    z = torch.linspace(0, 1, L, device=line.device)
    plane[0, c] = amplitude
    line[0, c, :, 0] = torch.exp(-((z - axis_frac) ** 2) / (2 * thickness * thickness))


@torch.no_grad()
def bake_stairs(tensorf, n_steps, z_near=0.55, z_far=0.93, thickness=0.012, amplitude=40.0, first_component=None, seed=0):
    """A back wall whose DEPTH varies across the picture: `n_steps` opaque slabs, step k covering one vertical strip of the
    box's x extent (in a shuffled order, so that neighbouring strips differ by several steps) at its own z fraction between
    z_near and z_far (rank-1 each: strip indicator on plane 0 times a bump on line 0).  A single fronto-parallel wall fills
    most of a forward-facing picture with content at ONE depth, for which a sideways camera translation and a small rotation
    are the same image motion -- the joint optimisation then explains the views by rotating the cameras
    (profiles/round3_llff_synthetic_pose_ambiguity.txt); depth steps across the picture make the two distinguishable."""
    plane, line = tensorf.density_plane[0], tensorf.density_line[0]
    C, W, L = plane.shape[1], plane.shape[3], line.shape[2]
    c0 = C - 1 - n_steps if first_component is None else first_component
    z = torch.linspace(0, 1, L, device=line.device)
    x = torch.linspace(0, 1, W, device=plane.device)
    order = np.random.RandomState(seed + 5).permutation(n_steps)
    for k in range(n_steps):
        lo, hi = k / n_steps, (k + 1) / n_steps
        strip = ((x >= lo) & ((x < hi) | (k == n_steps - 1))).float()
        zf = z_near + (z_far - z_near) * (order[k] + 0.5) / n_steps
        plane[0, c0 + k] = amplitude * strip[None, :].expand(plane.shape[2], -1)
        line[0, c0 + k, :, 0] = torch.exp(-((z - zf) ** 2) / (2 * thickness * thickness))


def make_gt_scene(opt, seed=0, res=None, n_blobs=12):
    """A ground-truth field for `opt`'s configuration: the same scene class, box, ranks and MLP as the run trains, at a
    grid of `res`^3-equivalent voxels (default opt.data.gt_res or 128), with opaque blobs (bake_blobs; an extra back wall
    for forward-facing NDC scenes), smooth textured appearance (bake_appearance) and the blur switched off (progress 1).
    Returns (gt_opt, gt_graph); render supervising views from it with render_views."""
    import copy
    from .model import bat_hip
    g_opt = copy.deepcopy(opt)
    res = int(res or (opt.data.get("gt_res", None) or 128))
    g_opt.train_schedule.n_voxel_init = res ** 3
    g_opt.train_schedule.resolution_scale_init = [1.0, 1.0, 1.0]
    g_opt.train_schedule.upsample_iters = []
    g_opt.train_schedule.n_voxel_final = res ** 3
    # building the field must not move the caller's random streams (same initial parameters with or without a GT scene)
    cpu_state = torch.get_rng_state()
    dev_state = torch.cuda.get_rng_state(torch.device(opt.device)) if str(opt.device).startswith("cuda") else None
    torch.manual_seed(90000 + seed)
    try:
        graph = bat_hip.Graph(g_opt).to(opt.device)
    finally:
        torch.set_rng_state(cpu_state)
        if dev_state is not None:
            torch.cuda.set_rng_state(dev_state, torch.device(opt.device))
    tf = graph.nerf.tensorf
    ndc = bool(opt.camera.ndc)
    if ndc:
        # NDC depth 1 - 2 n / z: real forward-facing content sits at z_ndc in [0.2, 0.9] (2.5 ... 20 near-plane distances);
        # the box's z runs from -2 to 1, so that is the far 27 % of it; a textured wall closes the scene behind the blobs
        stairs = int(opt.data.get("gt_stairs", None) or 0)
        C = tf.density_plane[0].shape[1]
        rad = tuple(opt.data.get("gt_blob_radius", None) or (0.08, 0.2))
        n = bake_blobs(tf, n_blobs=min(int(opt.data.get("gt_blobs", None) or n_blobs), C - 2 - stairs), seed=seed, radius=rad,
                       z_range=tuple(opt.data.get("gt_z_range", None) or (0.74, 0.93)))
        if stairs:   # a back wall in depth steps across the picture (bake_stairs) instead of one fronto-parallel slab
            bake_stairs(tf, stairs, z_near=float(opt.data.get("gt_stairs_near", None) or 0.55),
                        z_far=float(opt.data.get("gt_wall", None) or 0.93), seed=seed)
        else:
            bake_wall(tf, axis_frac=float(opt.data.get("gt_wall", None) or 0.975), thickness=0.012)
    else:
        n = bake_blobs(tf, n_blobs=min(n_blobs, tf.density_plane[0].shape[1] - 1), seed=seed)
    bake_appearance(tf, seed=seed)
    graph.nerf.set_progress(1.0)
    graph.eval()
    for p in graph.parameters():
        p.requires_grad_(False)
    graph.n_blobs = n
    return g_opt, graph


@torch.no_grad()
def render_views(g_opt, graph, views, chunk=8):
    """images [N,3,H,W] of the ground-truth field at views.pose / views.intr (the HIP evaluation renderer, mode "vis":
    no jitter, no blur, the yaml's background)."""
    out = []
    n = len(views.idx)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        ret = graph.render_by_slices(g_opt, views.pose[a:b], intr_inv=views.intr_inv[a:b], mode="vis", intr=views.intr[a:b])
        out.append(ret.rgb.view(b - a, g_opt.H, g_opt.W, 3).permute(0, 3, 1, 2).contiguous())
    return torch.cat(out, 0)


_SCENES = {}


def gt_scene_for(opt, seed=0):
    """the ground-truth field of (configuration, image size, seed), built once per process"""
    key = (str(opt.get("yaml", "")), str(opt.device), int(opt.H), int(opt.W), int(seed), int(opt.data.get("gt_res", None) or 128))
    if key not in _SCENES:
        _SCENES[key] = make_gt_scene(opt, seed=seed)
    return _SCENES[key]


def make_rendered_views(opt, n_views, seed=0, device="cuda", scene_seed=None):
    """make_views with the images rendered from the ground-truth field at the views' own (ground-truth) poses."""
    var = make_views(opt, n_views, seed=seed, device=device, with_images=False)
    g_opt, graph = gt_scene_for(opt, seed=int(opt.get("seed", 0)) if scene_seed is None else scene_seed)
    var.image = render_views(g_opt, graph, var)
    var.train_edge_masks = torch.zeros(n_views, opt.H * opt.W, dtype=torch.uint8, device=device)  # rebuilt by the 2-D cache
    return var


@torch.no_grad()
def load_scene_into(tensorf, gt_tensorf):
    """Resample the ground-truth field's factors onto `tensorf`'s grid (bilinear, align_corners, like the reference's own
    up_sampling_VM, tensoRF.py:274-295) and copy basis + MLP: a model that HAS converged to the scene -- the workload of
    the sharp last stage of a real run (few per cent of the in-box samples shaded), without training for it."""
    F = torch.nn.functional
    for name in ("density_plane", "app_plane"):
        for p, q in zip(getattr(tensorf, name), getattr(gt_tensorf, name)):
            p.copy_(F.interpolate(q, size=p.shape[2:], mode="bilinear", align_corners=True))
    for name in ("density_line", "app_line"):
        for p, q in zip(getattr(tensorf, name), getattr(gt_tensorf, name)):
            p.copy_(F.interpolate(q, size=p.shape[2:], mode="bilinear", align_corners=True))
    tensorf.basis_mat.weight.copy_(gt_tensorf.basis_mat.weight)
    for a, b in zip(tensorf.renderModule.weights(), gt_tensorf.renderModule.weights()):
        a.copy_(b)
