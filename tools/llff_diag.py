#!/usr/bin/env python3
"""What the joint optimisation recovers on the forward-facing synthetic scene (bat_llff_VM_MLP, schedule compressed 10 x): the
training images are fitted (loss 6e-5) by cameras that are NOT a similarity transform of the ground truth -- rotations of 1-12
degrees where the true cameras only translate: with a baseline of 0.3 against content at depth 4-40 the parallax is a few
pixels, and rotation stands in for translation.  The Procrustes alignment of the centres (model/bat.py:211-237) is then
meaningless (it warns that its SVD did not converge) and the held-out views rendered at aligned poses miss.  Prints centres,
rotation errors relative to view 0 and the residual of a scale-only fit.     python tools/llff_diag.py"""
import sys, json, numpy as np, torch
sys.path.insert(0, '.')
sys.argv = ['x']
import tools.converge as C
import argparse
args = argparse.Namespace(config='bat_llff_VM_MLP', compress=10.0, image_size=240, views=18, test_views=4, gt_res=128,
                          n_voxel_final=0, n_rays=0, noise=None, max_iter=0, test_iter=0, seed=0, graph=True,
                          report_every=0, llff_baseline=0.0, llff_focus=0.0)
torch.cuda.set_device(0)
opt, model = C.build(args)
model.train(opt)
pose, pose_gt = model.get_all_training_poses(opt)
np.set_printoptions(precision=4, suppress=True, linewidth=200)
def centers(p):
    R, t = p[:, :, :3], p[:, :, 3:]
    return (-R.transpose(1, 2) @ t)[..., 0]
c, cg = centers(pose).cpu().numpy(), centers(pose_gt).cpu().numpy()
print("GT centres\n", cg[:8]); print("pred centres\n", c[:8])
print("GT centre std", cg.std(0), "pred centre std", c.std(0))
# relative rotation error without any alignment: R_i R_0^T vs GT
R, Rg = pose[:, :, :3].cpu().numpy(), pose_gt[:, :, :3].cpu().numpy()
def ang(A):
    return np.degrees(np.arccos(np.clip((np.trace(A) - 1) / 2, -1, 1)))
rel = [ang((R[i] @ R[0].T) @ (Rg[i] @ Rg[0].T).T) for i in range(len(R))]
print("relative rotation error vs view 0 (deg):", np.round(rel, 3))
print("abs rotation of pred vs identity (deg):", np.round([ang(R[i]) for i in range(len(R))], 3))
print("abs rotation of GT vs identity (deg):", np.round([ang(Rg[i]) for i in range(len(R))], 3))
# best similarity between centre sets (least squares scale only, no rotation)
cc, cgc = c - c.mean(0), cg - cg.mean(0)
s = (cc * cgc).sum() / (cc * cc).sum()
print("scale pred->gt (no rotation)", s, "residual", np.abs(s * cc - cgc).max(), "gt extent", np.abs(cgc).max())
