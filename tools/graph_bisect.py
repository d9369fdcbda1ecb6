#!/usr/bin/env python3
"""Debug helper: capture growing prefixes of the training iteration of the small test model into a hipGraph,
each in its own process (a crash inside hipStreamEndCapture must not take the others down)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) == 2:
    what = sys.argv[1]
    import faulthandler; faulthandler.enable()
    import numpy as np, torch
    import test_gpu_graph as T
    from joint_tensorf_amd.options import Opt
    if what.startswith("multi"):
        from joint_tensorf_amd.graphed import GraphedTrainStep
        opt, model, var0 = T._build()
        model.it = 9000  # no edge loss: one signature per lattice shape
        np.random.seed(5)
        np.random.randint = lambda *a, **k: 0
        st = GraphedTrainStep(model, min_repeats=0)
        if what == "multi_nopool":
            GraphedTrainStep._capture_orig = GraphedTrainStep._capture
            def cap(self, *a, **k):
                self.pool = None
                return GraphedTrainStep._capture_orig(self, *a, **k)
            GraphedTrainStep._capture = cap
        model.train_iteration(opt, Opt(dict(var0)))   # eager: workspaces
        def run(n, tag):
            for i in range(n):
                loss = st.train_iteration(opt, Opt(dict(var0)))
                model.after_iteration(opt, model.it - 1)
                print(tag, i, st.stats, float(loss.all.detach()), flush=True)
        run(4, "A")
        np.random.randint = lambda *a, **k: 7      # other lattice shape -> second graph
        run(3, "B")
        np.random.randint = lambda *a, **k: 0
        run(3, "A")
        np.random.randint = lambda *a, **k: 7
        run(3, "B")
        sys.exit(0)
    if what.startswith("edge"):
        from joint_tensorf_amd.graphed import GraphedTrainStep
        opt, model, var0 = T._build()
        if what == "edge_off":
            opt.edge_mask_on_render_loss = False
        if what == "edge_always":
            opt.alternate_edge_loss = False
        np.random.seed(5)
        np.random.randint = lambda *a, **k: 0
        st = GraphedTrainStep(model, min_repeats=0)
        model.train_iteration(opt, Opt(dict(var0)))   # eager: workspaces
        tf = model.graph.nerf.tensorf
        for i in range(8):
            loss = st.train_iteration(opt, Opt(dict(var0)))
            model.after_iteration(opt, model.it - 1)
            print(i, st.stats, "loss %.5f render %.5f L1 %.5f" % (float(loss.all.detach()), float(loss.render.detach()), float(loss.L1.detach())),
                  "pmax %.4f amax %.4f w %.4f se3 %.5f" % (float(tf.density_plane[0].abs().max()), float(tf.app_plane[0].abs().max()),
                  float(tf.renderModule.weights()[0].abs().max()), float(model.graph.se3_refine.weight.abs().max())), flush=True)
        sys.exit(0)
    if what.startswith("stepper"):
        from joint_tensorf_amd.graphed import GraphedTrainStep
        opt, model, var0 = T._build()
        np.random.seed(5)
        if what.startswith("stepper0"):
            np.random.randint = lambda *a, **k: 0
        st = GraphedTrainStep(model, min_repeats=1)
        tf = model.graph.nerf.tensorf
        loss = None
        for i in range(9):
            if what == "stepper0_drop":
                loss = None
                st.last_var = None
            loss = st.train_iteration(opt, Opt(dict(var0)))
            model.after_iteration(opt, model.it - 1)
            torch.cuda.synchronize()
            print(i, st.stats, "loss %.5g render %.5g L1 %.5g" % (float(loss.all.detach()), float(loss.render.detach()), float(loss.L1.detach())),
                  "pmax %.4g amax %.4g w %.4g se3 %.5g" % (float(tf.density_plane[0].abs().max()), float(tf.app_plane[0].abs().max()),
                  float(tf.renderModule.weights()[0].abs().max()), float(model.graph.se3_refine.weight.abs().max())), flush=True)
        sys.exit(0)
    opt, model, var0 = T._build()
    it0 = int(os.environ.get("IT0", "0"))
    model.it = it0
    g = model.graph
    np.random.seed(5)
    np.random.randint = lambda *a, **k: 0   # densest lattice everywhere: workspaces are final after step 1
    for _ in range(2):
        model.train_iteration(opt, Opt(dict(var0)))
    g.it = model.it
    torch.cuda.synchronize()
    model.optim.zero_grad(); model.optim_pose.zero_grad()
    G = torch.cuda.CUDAGraph()
    err = None
    with torch.cuda.graph(G):
      try:
        if what == "lat":
            off = torch.zeros(2, device="cuda", dtype=torch.int32)
            bx = torch.arange(5, device="cuda") * 8
            def lattice(_s):
                o = off.long()
                sx, sy = bx + o[0], bx + o[1]
                return (sx[None, :] + sy[:, None] * 42).reshape(-1), 5, 5
            g.lattice_override = lattice
            v = g.forward(opt, Opt(dict(var0)), mode="train")
            loss = g.compute_loss(opt, v, mode="train")
            loss = model.summarize_loss(opt, v, loss)
            loss.all.backward()
            model.optim.launch_step()
        elif what == "pose":
            pose = g.get_pose(opt, Opt(dict(var0)), mode="train")
        elif what == "fwd_nograd":
            with torch.no_grad():
                v = g.forward(opt, Opt(dict(var0)), mode="train")
        else:
            v = g.forward(opt, Opt(dict(var0)), mode="train")
            if what != "fwd":
                loss = g.compute_loss(opt, v, mode="train")
                loss = model.summarize_loss(opt, v, loss)
                if what == "bwd_render":
                    loss.render.backward()
                elif what == "bwd_l1":
                    loss.L1.backward()
                elif what != "loss":
                    loss.all.backward()
                    if what == "all":
                        model.optim.launch_step()
      except Exception as ex:
        err = ex
    if err is not None:
        raise err
    print(what, "captured", flush=True)
    G.replay(); torch.cuda.synchronize()
    print(what, "replayed", flush=True)
    sys.exit(0)

for what in sys.argv[2:] if len(sys.argv) > 2 else ("stepper0", "stepper0_drop"):
    pass

if len(sys.argv) != 2:
    names = sys.argv[2:] if len(sys.argv) > 2 else ["stepper0", "stepper0_drop"]
    for what in names:
        r = subprocess.run([sys.executable, __file__, what], capture_output=True, text=True)
        lines = [l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l and not l.startswith("  File")]
        print("==", what, "rc", r.returncode)
        print("\n".join(l[:300] for l in lines[:40]), flush=True)
