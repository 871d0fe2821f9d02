"""Second, independent restatement of the reference's depth -> flow geometry.  TEST INFRASTRUCTURE ONLY.

Follows multivih5datareaderop/multivih5datareader.cpp:369-424 (`computeFlow`, the ground-truth flow the training reader
produces) and :431-501 (`computeDepthmask`), which are written in WORLD coordinates with two absolute cameras -- unlike
oracle/demon_oracle.c:ref_depth_to_flow and the lmbspecialops op, which take the relative motion of camera 2 w.r.t. camera 1:

    K_i   = diag-scaled intrinsics (fx*W, fy*H, cx*W, cy*H)                       (:373-388)
    pos   = K1^-1 (x+0.5, y+0.5, 1) * depth / norm     (norm = |ray| for RAY_LENGTH depth, else 1)   (:398-414)
    pos   = R1^T (pos - t1)                              camera 1 -> world          (:416-417)
    p2    = K2 [R2|t2] (pos, 1);  p2.xy /= p2.z                                     (:419-421)
    flow  = p2.xy - (x+0.5, y+0.5);  NaN where depth <= 0 or not finite            (:404-409, :423-424)

The file is a TensorFlow op with HDF5 / Eigen dependencies and cannot be compiled here, hence a restatement (float32 numpy, same
operation order).  tests/test_oracle.py checks that it agrees with (a) the golden flow produced by the reference's Cython routine
(tests/golden/sculpture_geometry.npz: flow12) and (b) the oracle's depth_to_flow fed with the relative motion
R = R2 R1^T, t = t2 - R t1 -- two restatements of different reference files, written in different frames, and one piece of
reference code run here all have to agree.
"""
import numpy as np

f32 = np.float32


def _K(intrinsics, W, H):
    fx, fy, cx, cy = (f32(v) for v in intrinsics)
    K = np.zeros((3, 3), f32)
    K[0, 0] = fx * f32(W)
    K[1, 1] = fy * f32(H)
    K[0, 2] = cx * f32(W)
    K[1, 2] = cy * f32(H)
    K[2, 2] = 1
    return K


def _project(depth, intr1, R1, t1, intr2, R2, t2, ray_length):
    H, W = depth.shape
    K2 = _K(intr2, W, H)
    P2 = (K2 @ np.concatenate([np.asarray(R2, f32), np.asarray(t2, f32).reshape(3, 1)], axis=1)).astype(f32)
    inv_K = np.linalg.inv(_K(intr1, W, H).astype(np.float64)).astype(f32)
    inv_R = np.asarray(R1, f32).T
    t = np.asarray(t1, f32)
    y, x = np.mgrid[0:H, 0:W]
    p1x, p1y = x.astype(f32) + f32(0.5), y.astype(f32) + f32(0.5)
    pos = np.stack([inv_K[0, 0] * p1x + inv_K[0, 2], inv_K[1, 1] * p1y + inv_K[1, 2], np.ones_like(p1x)], axis=0)
    norm = np.sqrt((pos * pos).sum(axis=0)) if ray_length else f32(1)
    d = np.asarray(depth, f32)
    valid = np.isfinite(d) & (d > 0)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        pos = pos * (d / norm)
        pos = pos - t[:, None, None]
        pos = np.einsum("ij,jhw->ihw", inv_R, pos).astype(f32)
        p2 = (np.einsum("ij,jhw->ihw", P2[:, :3], pos) + P2[:, 3][:, None, None]).astype(f32)
        p2x, p2y = p2[0] / p2[2], p2[1] / p2[2]
    return p1x, p1y, p2x, p2y, valid


def compute_flow(depth, intr1, R1, t1, intr2, R2, t2, ray_length=False):
    """multivih5datareader.cpp:369-424 -> [2,H,W] flow in pixels from the image of cam1 to the image of cam2"""
    p1x, p1y, p2x, p2y, valid = _project(depth, intr1, R1, t1, intr2, R2, t2, ray_length)
    flow = np.stack([p2x - p1x, p2y - p1y], axis=0).astype(f32)
    flow[:, ~valid] = np.nan
    return flow


def compute_depthmask(depth, intr1, R1, t1, intr2, R2, t2, border1=0, border2=0, ray_length=False):
    """multivih5datareader.cpp:431-501 -> [H,W] uint8, 1 where the point is inside both images (minus the borders)"""
    H, W = depth.shape
    p1x, p1y, p2x, p2y, valid = _project(depth, intr1, R1, t1, intr2, R2, t2, ray_length)
    y, x = np.mgrid[0:H, 0:W]
    inside1 = (x >= border1) & (y >= border1) & (x < W - border1) & (y < H - border1)
    with np.errstate(invalid="ignore"):
        inside2 = ~((p2x < border2) | (p2y < border2) | (p2x >= W - border2) | (p2y >= H - border2))
    return (inside1 & valid & inside2).astype(np.uint8)
