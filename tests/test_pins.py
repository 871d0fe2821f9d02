"""CPU tests that pin the oracle to reference-held material beyond tests/test_oracle.py (round 2):

  * a second, independent restatement of the reference's flow-from-depth (multivih5datareader.cpp:369-424, world frame,
    oracle/reader_ref.py) agrees with the golden flow computed by the reference's Cython routine and with the oracle's
    depth_to_flow fed with the relative motion;
  * photometric known-answer test: warping sculpture image 2 by the reference flow (oracle warp2d) resembles image 1 on
    the reference's visible-pixel mask -- pins sign, channel order and normalisation of warp2d's displacement with
    reference-held images (SURVEY.md section 8c golden use (2));
  * semantics decided here: NaN ordering of median3x3_downsample, border rule of scale_invariant_gradient (its layout --
    channels folded into the batch, deltas summed -- follows the lmbspecialops contract), depth_to_normals.
"""
import os

import numpy as np
import pytest

from oracle import ops_ref, reader_ref

HERE = os.path.dirname(os.path.abspath(__file__))
K_DEMON = np.array([0.89115971, 1.18821287, 0.5, 0.5], np.float32)


def golden():
    g = np.load(os.path.join(HERE, "golden", "sculpture_geometry.npz"))
    i = np.load(os.path.join(HERE, "golden", "sculpture_inputs.npz"))
    img1 = (i["image1_u8"].astype(np.float32) / 255 - 0.5).transpose(2, 0, 1)[None]
    img2 = (i["image2_u8"].astype(np.float32) / 255 - 0.5).transpose(2, 0, 1)[None]
    return g, np.ascontiguousarray(img1), np.ascontiguousarray(img2)


def aa_from_R(R):
    angle = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    axis = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(angle))
    return (axis * angle).astype(np.float32)


def ncc(a, b, mask):
    """normalised cross correlation of the grey values on the masked pixels"""
    a = a[0].mean(axis=0)[mask].astype(np.float64)
    b = b[0].mean(axis=0)[mask].astype(np.float64)
    a -= a.mean()
    b -= b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


def photometric_scores(warp, g, img1, img2):
    """warp(img, displacement, normalized, border_mode) -> scores of the known-answer test (shared with the GPU test)"""
    H, W = g["depth1"].shape
    flow = np.where(np.isfinite(g["flow12"]), g["flow12"], 0).astype(np.float32)[None]
    mask = g["visible_mask"].astype(bool) & (np.abs(g["depth_ratios"] - 1) < 0.02)   # visible and depth-consistent
    norm = flow / np.array([W, H], np.float32)[None, :, None, None]
    return {
        "pixels": ncc(warp(img2, flow, False, "value"), img1, mask),
        "normalized": ncc(warp(img2, norm, True, "value"), img1, mask),
        "unwarped": ncc(img2, img1, mask),
        "negated": ncc(warp(img2, -flow, False, "value"), img1, mask),
        "swapped": ncc(warp(img2, np.ascontiguousarray(flow[:, ::-1]), False, "value"), img1, mask),
        "pixel_flow_as_normalized": ncc(warp(img2, flow, True, "value"), img1, mask),
        "mask_fraction": float(mask.mean()),
    }


def check_photometric(s):
    assert s["mask_fraction"] > 0.3
    assert s["pixels"] > 0.6                                  # image 2 pulled back by the reference flow looks like image 1
    assert abs(s["normalized"] - s["pixels"]) < 1e-3          # normalized=True means flow / (W, H)
    assert s["unwarped"] < 0.35 and s["negated"] < 0.2 and s["swapped"] < 0.35   # wrong sign / channel order fail clearly
    assert s["pixel_flow_as_normalized"] < 0.2                # a missing (W, H) factor fails clearly


def test_photometric_warp_kat_oracle():
    g, img1, img2 = golden()
    check_photometric(photometric_scores(ops_ref.warp2d, g, img1, img2))


def test_reader_flow_restatement_agrees_with_golden_and_oracle():
    g, _, _ = golden()
    depth1, Rt1, Rt2 = g["depth1"], g["Rt1"], g["Rt2"]
    gold = g["flow12"]
    f = reader_ref.compute_flow(depth1, K_DEMON, Rt1[:, :3], Rt1[:, 3], K_DEMON, Rt2[:, :3], Rt2[:, 3])
    assert np.array_equal(np.isnan(f), np.isnan(gold))
    m = np.isfinite(gold)
    assert np.abs(f[m] - gold[m]).max() < 1e-3               # two pieces of reference code, one run here, one restated
    # world-frame invariance: moving both cameras by the same rigid transform G (R_i' = R_i G^T, t_i' = t_i - R_i' c) leaves the
    # flow unchanged; this exercises the camera-1 -> world step (:416-417) that the identity Rt1 of the fixture hides
    rng = np.random.default_rng(3)
    A = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    G = A * np.sign(np.linalg.det(A))
    c = rng.standard_normal(3)
    cams = []
    for Rt in (Rt1, Rt2):
        R = Rt[:, :3] @ G.T
        cams.append((R, Rt[:, 3] - R @ c))
    f2 = reader_ref.compute_flow(depth1, K_DEMON, cams[0][0], cams[0][1], K_DEMON, cams[1][0], cams[1][1])
    assert np.abs(f2[m] - gold[m]).max() < 2e-3
    # relative motion into the oracle's op: X2 = R2 (R1^T (X1 - t1)) + t2 = R X1 + t
    R = cams[1][0] @ cams[0][0].T
    t = cams[1][1] - R @ cams[0][1]
    o = ops_ref.depth_to_flow(depth1[None, None], K_DEMON, aa_from_R(R)[None], t.astype(np.float32)[None], False, False)[0]
    assert np.array_equal(np.isnan(o), np.isnan(f2))
    assert np.abs(o[m] - f2[m]).max() < 2e-3
    # ray-length depth (Camera::RAY_LENGTH, :411-412) == z depth of the same points
    H, W = depth1.shape
    ys, xs = np.mgrid[0:H, 0:W]
    ray = np.sqrt(((xs + 0.5 - K_DEMON[2] * W) / (K_DEMON[0] * W)) ** 2 + ((ys + 0.5 - K_DEMON[3] * H) / (K_DEMON[1] * H)) ** 2 + 1)
    f3 = reader_ref.compute_flow((depth1 * ray).astype(np.float32), K_DEMON, Rt1[:, :3], Rt1[:, 3], K_DEMON, Rt2[:, :3], Rt2[:, 3], ray_length=True)
    assert np.abs(f3[m] - gold[m]).max() < 2e-3
    # computeDepthmask (:431-501) with zero borders == the reference's visible mask restricted to "inside both images"
    dm = reader_ref.compute_depthmask(depth1, K_DEMON, Rt1[:, :3], Rt1[:, 3], K_DEMON, Rt2[:, :3], Rt2[:, 3])
    vis = g["visible_mask"].astype(bool)
    assert not (vis & ~dm.astype(bool)).any()                 # every visible point lies inside both images
    assert dm[~m[0]].sum() == 0                               # invalid depth is never in the mask


def test_median_nan_sorts_last():
    x = np.arange(25, dtype=np.float32).reshape(1, 1, 5, 5)
    base = ops_ref.median3x3_downsample(x)
    for k in range(1, 5):   # up to four NaNs in the window of output (1,1) (centre (2,2)): still a number
        y = x.copy()
        idx = [(1, 1), (1, 2), (1, 3), (2, 1)][:k]
        for (r, c) in idx:
            y[0, 0, r, c] = np.nan
        m = ops_ref.median3x3_downsample(y)
        win = np.sort(np.where(np.isnan(y[0, 0, 1:4, 1:4]), np.inf, y[0, 0, 1:4, 1:4]).reshape(-1))
        assert m[0, 0, 1, 1] == win[4] and np.isfinite(m[0, 0, 1, 1])
    y = x.copy()
    for (r, c) in [(1, 1), (1, 2), (1, 3), (2, 1), (2, 2)]:
        y[0, 0, r, c] = np.nan
    assert np.isnan(ops_ref.median3x3_downsample(y)[0, 0, 1, 1])   # five NaNs: the median is NaN
    y = x.copy()
    y[0, 0, 3, 3] = np.inf     # the largest and the smallest of the window
    y[0, 0, 1, 1] = -np.inf
    assert ops_ref.median3x3_downsample(y)[0, 0, 1, 1] == base[0, 0, 1, 1]   # +-inf are ordinary ordered values


def test_sig_layout_for_channels_and_deltas():
    """lmbspecialops contract: [N,C,H,W] -> [N*C,2,H,W], the deltas of one call are summed with their weights;
    one delta per call + concat on axis 1 (v2/losses.py:76-79) gives the (x, y) pairs the loss slices (:99-102); C = 2 is the
    flow case (:343) and yields a batch of 2N"""
    rng = np.random.default_rng(0)
    u = rng.standard_normal((2, 2, 9, 11)).astype(np.float32)
    deltas, weights = [1, 2, 4], [1.0, 0.5, 2.0]
    full = ops_ref.scale_invariant_gradient(u, deltas, weights, 0.01)
    assert full.shape == (2 * 2, 2, 9, 11)
    total = np.zeros_like(full)
    for d, w in zip(deltas, weights):
        single = ops_ref.scale_invariant_gradient(u, [d], [w], 0.01)
        assert single.shape == (4, 2, 9, 11)
        for b in range(2):
            for c in range(2):   # row b*C + c of the folded batch is channel c of sample b on its own
                np.testing.assert_array_equal(single[b * 2 + c], ops_ref.scale_invariant_gradient(u[b:b + 1, c:c + 1], [d], [w], 0.01)[0])
        total += single
    np.testing.assert_allclose(full, total, atol=1e-6)
    # direct formula, delta 2, x direction, interior and right border
    s = ops_ref.scale_invariant_gradient(u[:, :1], [2], [0.5], 0.01)
    a, b = u[0, 0, 3, 4], u[0, 0, 3, 6]
    np.testing.assert_allclose(s[0, 0, 3, 4], 0.5 * (b - a) / (abs(a) + abs(b) + 0.01), rtol=1e-6)
    assert s[0, 0, 3, 9] == 0 and s[0, 0, 3, 10] == 0 and s[0, 1, 7, 0] == 0 and s[0, 1, 8, 0] == 0


def test_flow_sig_loss_layout_matches_the_reference_call_site():
    """v2/losses.py:176-177 on a C = 2 tensor: scale_invariant_gradient(flow) is [2N, 10, H, W] and loss_flow2_sig is ONE
    pointwise_l2_loss over those 10 channels (sqrt of the sum over 10 channels, mean over 2N*H*W) -- not 20 channels over N"""
    rng = np.random.default_rng(1)
    n, h, w = 2, 6, 8
    flow = rng.standard_normal((n, 2, h, w)).astype(np.float32)
    gt = rng.standard_normal((n, 2, h, w)).astype(np.float32)
    deltas = [1, 2, 4]
    sig = lambda t: np.concatenate([ops_ref.scale_invariant_gradient(t, [d], [1.0], 0.001) for d in deltas], axis=1)
    a, b = sig(flow), sig(gt)
    assert a.shape == (2 * n, 2 * len(deltas), h, w)
    want = np.sqrt(((a.astype(np.float64) - b) ** 2).sum(axis=1) + 1e-3).mean()
    got = ops_ref.pointwise_l2_loss(a, b, 1e-3)
    assert abs(got - want) < 1e-5 * want


def plane_depth(n, d, H, W, K=K_DEMON):
    """z-depth map of the plane n . X = d seen through intrinsics K"""
    ys, xs = np.mgrid[0:H, 0:W]
    rx = (xs + 0.5 - K[2] * W) / (K[0] * W)
    ry = (ys + 0.5 - K[3] * H) / (K[1] * H)
    return (d / (n[0] * rx + n[1] * ry + n[2])).astype(np.float32)


@pytest.mark.parametrize("inverse", [False, True])
def test_depth_to_normals_planes(inverse):
    H, W = 24, 32
    for nrm in ([0, 0, -1.0], [0.3, -0.2, -0.9], [-0.5, 0.4, -0.7]):
        nrm = np.array(nrm) / np.linalg.norm(nrm)
        z = plane_depth(nrm, nrm[2] * 2.0, H, W)     # passes through (0, 0, 2)
        assert (z > 0).all()
        inp = (1 / z if inverse else z)[None, None]
        out = ops_ref.depth_to_normals(inp, K_DEMON, inverse_depth=inverse)
        assert out.shape == (1, 3, H, W)
        assert np.isnan(out[0, :, 0]).all() and np.isnan(out[0, :, -1]).all() and np.isnan(out[0, :, :, 0]).all() and np.isnan(out[0, :, :, -1]).all()
        inner = out[0, :, 1:-1, 1:-1]
        np.testing.assert_allclose(inner, np.broadcast_to(nrm[:, None, None], inner.shape), atol=2e-4)   # towards the camera
        np.testing.assert_allclose(np.linalg.norm(inner, axis=0), 1, atol=1e-5)
    # invalid depth poisons the pixel and its four neighbours only
    z = plane_depth(np.array([0, 0, -1.0]), -2.0, H, W)
    z[10, 10] = 0
    z[5, 20] = np.nan
    out = ops_ref.depth_to_normals(z[None, None], K_DEMON)[0]
    bad = np.isnan(out[0, 1:-1, 1:-1])
    want = np.zeros((H, W), bool)
    for (y, x) in ((10, 10), (5, 20)):
        for (dy, dx) in ((0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)):
            want[y + dy, x + dx] = True
    assert np.array_equal(bad, want[1:-1, 1:-1])
    # a depth step: the one-sided difference with the smaller depth change keeps both sides' normals clean
    z = np.full((H, W), 2.0, np.float32)
    z[:, 16:] = 3.0
    out = ops_ref.depth_to_normals(z[None, None], K_DEMON)[0]
    np.testing.assert_allclose(out[:, 5, 15], [0, 0, -1], atol=1e-5)
    np.testing.assert_allclose(out[:, 5, 16], [0, 0, -1], atol=1e-5)
