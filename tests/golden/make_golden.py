"""Generates tests/golden/sculpture_geometry.npz by RUNNING THE REFERENCE'S OWN CODE in this container.

  inputs : /root/reference/examples/sculpture_depth{1,2}.npy, sculpture_Rt2.txt (the only fixtures the
           reference ships) and the normalised intrinsics of examples/example.py:52-60
  code   : reference python/depthmotionnet/dataset_tools/view_tools_cython.pyx, functions
           compute_visible_points_mask (:62-102) and compute_depth_ratios (:164-191), built by
           oracle/build_ref.py into oracle/_ref/
           and the module-private _compute_flow (:196-244) through oracle/build_ref.py build_flow()
  outputs: visible mask, depth ratios and optical flow of view 1 projected into view 2

These pin the depth -> flow geometry convention of the oracle (tests/test_oracle.py::test_golden_*).
Run:  python tests/golden/make_golden.py      (needs /root/reference; the GPU box only uses the .npz)
"""
import collections
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import build_ref  # noqa: E402

View = collections.namedtuple("View", ["R", "t", "K", "image", "depth", "depth_metric"])  # dataset_tools/view.py:25


def main():
    so = build_ref.build()
    if so is None:
        raise SystemExit("reference tree not available")
    spec = importlib.util.spec_from_file_location("view_tools_cython", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    ex = os.path.join(build_ref.REF, "examples")
    depth1 = np.load(os.path.join(ex, "sculpture_depth1.npy")).astype(np.float32)
    depth2 = np.load(os.path.join(ex, "sculpture_depth2.npy")).astype(np.float32)
    Rt1 = np.loadtxt(os.path.join(ex, "sculpture_Rt1.txt"))
    Rt2 = np.loadtxt(os.path.join(ex, "sculpture_Rt2.txt"))
    H, W = depth1.shape
    intr = np.array([0.89115971, 1.18821287, 0.5, 0.5])
    K = np.array([[intr[0] * W, 0, intr[2] * W], [0, intr[1] * H, intr[3] * H], [0, 0, 1]], np.float64)
    v1 = View(R=Rt1[:, :3], t=Rt1[:, 3], K=K, image=None, depth=depth1, depth_metric="camera_z")
    v2 = View(R=Rt2[:, :3], t=Rt2[:, 3], K=K, image=None, depth=depth2, depth_metric="camera_z")
    mask = np.asarray(mod.compute_visible_points_mask(v1, v2)).astype(np.uint8)
    ratios = np.asarray(mod.compute_depth_ratios(v1, v2)).astype(np.float32)
    # the reference's own flow-from-depth routine (view_tools_cython.pyx:196-244), reached through a wrapper that includes the
    # source where it lies (oracle/build_ref.py build_flow): flow of view 1 into view 2 in pixels, NaN where depth is invalid
    fso = build_ref.build_flow()
    fspec = importlib.util.spec_from_file_location("ref_flow", fso)
    fmod = importlib.util.module_from_spec(fspec)
    fspec.loader.exec_module(fmod)
    P2 = K @ Rt2
    flow12 = np.asarray(fmod.compute_flow(depth1, K, Rt1[:, :3], Rt1[:, 3], P2)).astype(np.float32)
    out = os.path.join(HERE, "sculpture_geometry.npz")
    np.savez_compressed(out, depth1=depth1, depth2=depth2, Rt1=Rt1, Rt2=Rt2, intrinsics=intr.astype(np.float32),
                        visible_mask=mask, depth_ratios=ratios, flow12=flow12)
    print(out, "mask visible %.3f" % mask.mean(), "ratio median %.4f" % np.nanmedian(ratios))


if __name__ == "__main__":
    main()
