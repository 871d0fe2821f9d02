"""GPU parity of the lmbspecialops-level HIP kernels against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

from conftest import rel_l1
from oracle import ops_ref

pytestmark = pytest.mark.gpu
K_DEMON = np.array([0.89115971, 1.18821287, 0.5, 0.5], np.float32)


def _motion(rng, n):
    rot = (rng.standard_normal((n, 3)) * 0.1).astype(np.float32)
    tr = rng.standard_normal((n, 3)).astype(np.float32)
    tr /= np.linalg.norm(tr, axis=1, keepdims=True)
    return rot, tr


@pytest.mark.parametrize("shape", [(1, 48, 64), (3, 48, 64), (2, 120, 160), (1, 5, 7)])
@pytest.mark.parametrize("inverse,normalize,gate", [(True, True, True), (False, False, False), (True, False, False)])
def test_depth_to_flow(gpu_ctx, shape, inverse, normalize, gate):
    n, h, w = shape
    rng = np.random.default_rng(10)
    d = (0.2 + rng.random((n, 1, h, w))).astype(np.float32)
    d.reshape(-1)[::17] = 0          # invalid depths -> NaN (-> 0 with the gate)
    d.reshape(-1)[::29] = -1
    d.reshape(-1)[::31] = np.nan
    rot, tr = _motion(rng, n)
    want = ops_ref.depth_to_flow(d, K_DEMON, rot, tr, inverse, normalize, gate)
    got = gpu_ctx.depth_to_flow(d, K_DEMON, rot, tr, inverse, normalize, gate)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    assert rel_l1(got[m], want[m]) < 1e-4


def _golden_sculpture():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sculpture_geometry.npz"))
    R = g["Rt2"][:, :3].astype(np.float64)
    angle = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    axis = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(angle))
    return g, (axis * angle).astype(np.float32)[None], g["Rt2"][:, 3].astype(np.float32)[None]


def test_depth_to_flow_matches_reference_golden(gpu_ctx):
    """HIP depth_to_flow == the flow the reference's own routine (view_tools_cython.pyx:196-244) computes for the sculpture
    pair (tests/golden/make_golden.py ran it in the build container): pixels, NaN at invalid depth"""
    g, aa, t = _golden_sculpture()
    depth1, want = g["depth1"], g["flow12"]
    got = gpu_ctx.depth_to_flow(depth1[None, None], K_DEMON, aa, t, False, False, False)[0]
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    m = np.isfinite(want)
    assert np.abs(got[m] - want[m]).max() < 2e-3 and rel_l1(got[m], want[m]) < 1e-5


@pytest.mark.parametrize("method", [0, 1])
def test_flow_to_depth_recovers_reference_depth(gpu_ctx, method):
    """flow_to_depth (DLT / SVD) and flow_to_depth2 (closed form) applied to the REFERENCE's flow with the sculpture motion
    give back the reference's depth map (depth, not inverse; flow in pixels)"""
    g, aa, t = _golden_sculpture()
    depth1, flow = g["depth1"], g["flow12"]
    m = np.isfinite(flow).all(0)
    f = np.where(m, flow, 0.0).astype(np.float32)
    got = gpu_ctx.flow_to_depth(f[None], K_DEMON, aa, t, False, False, method)[0, 0]
    assert rel_l1(got[m], depth1[m]) < 1e-4


@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("shape", [(1, 48, 64), (3, 48, 64), (2, 120, 160)])
def test_flow_to_depth(gpu_ctx, method, shape):
    n, h, w = shape
    rng = np.random.default_rng(11)
    inv_depth = (0.2 + rng.random((n, 1, h, w))).astype(np.float32)
    rot, tr = _motion(rng, n)
    flow = ops_ref.depth_to_flow(inv_depth, K_DEMON, rot, tr, True, True)
    flow += (rng.standard_normal(flow.shape) * 0.002).astype(np.float32)  # inconsistent flow, like a prediction
    want = ops_ref.flow_to_depth(flow, K_DEMON, rot, tr, True, True, method)
    got = gpu_ctx.flow_to_depth(flow, K_DEMON, rot, tr, True, True, method)
    assert rel_l1(got, want) < 1e-3
    # round trip property at full size: consistent flow gives the depth back
    back = gpu_ctx.flow_to_depth(ops_ref.depth_to_flow(inv_depth, K_DEMON, rot, tr, True, True), K_DEMON, rot, tr, True, True, method)
    assert rel_l1(back, inv_depth) < 2e-3


@pytest.mark.parametrize("shape", [(1, 3, 48, 64), (3, 3, 48, 64), (2, 3, 120, 160), (1, 1, 5, 7), (2, 5, 9, 130)])
@pytest.mark.parametrize("normalized,border", [(True, "value"), (False, "clamp"), (False, "value")])
def test_warp2d(gpu_ctx, shape, normalized, border):
    n, c, h, w = shape
    rng = np.random.default_rng(12)
    img = rng.random(shape).astype(np.float32)
    disp = rng.standard_normal((n, 2, h, w)).astype(np.float32) * (0.2 if normalized else 6.0)
    disp[0, 0, 0, 0] = np.nan
    disp[0, 1, h - 1, w - 1] = np.inf
    disp[0, :, h // 2, w // 2] = 0.0
    want = ops_ref.warp2d(img, disp, normalized, border, 0.25)
    got = gpu_ctx.warp2d(img, disp, normalized, border, 0.25)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    assert np.abs(got[m] - want[m]).max() < 1e-4  # floor() can only flip where a,b ~ 0 or 1: continuous
    # identity property
    zero = np.zeros((n, 2, h, w), np.float32)
    np.testing.assert_array_equal(gpu_ctx.warp2d(img, zero, normalized, border), img)


def test_elementwise_and_stencils(gpu_ctx):
    rng = np.random.default_rng(13)
    for count in (0, 1, 3, 4, 1023, 4096 + 5):
        x = rng.standard_normal(count).astype(np.float32)
        if count > 3:
            x[1], x[2], x[3] = np.nan, np.inf, -np.inf
        got, want = gpu_ctx.leaky_relu(x, 0.1), ops_ref.leaky_relu(x, 0.1)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(gpu_ctx.replace_nonfinite(x, 2.5), ops_ref.replace_nonfinite(x, 2.5))
    u = rng.standard_normal((2, 3, 48, 64)).astype(np.float32)
    for deltas, weights in (([1], [1.0]), ([2], [0.5]), ([1, 2, 4, 8, 16], [1, 0.5, 0.25, 0.125, 0.0625])):
        got = gpu_ctx.scale_invariant_gradient(u, deltas, weights, 0.01)
        want = ops_ref.scale_invariant_gradient(u, deltas, weights, 0.01)
        assert got.shape == want.shape == (2 * 3, 2, 48, 64)   # channels fold into the batch, deltas are summed
        assert np.abs(got - want).max() < 1e-5
    # C = 2 (the flow case, v2/losses.py:343), one delta per call as the reference does (:76-79)
    f = rng.standard_normal((3, 2, 48, 64)).astype(np.float32)
    for d in (1, 2, 4, 8, 16):
        got = gpu_ctx.scale_invariant_gradient(f, [d], [1.0], 0.001)
        assert got.shape == (6, 2, 48, 64)
        for b in range(3):
            for c in range(2):
                np.testing.assert_allclose(got[b * 2 + c], ops_ref.scale_invariant_gradient(f[b:b + 1, c:c + 1], [d], [1.0], 0.001)[0], atol=1e-5)
    for shape in ((2, 3, 192, 256), (1, 1, 7, 9)):
        x = rng.standard_normal(shape).astype(np.float32)
        np.testing.assert_array_equal(gpu_ctx.median3x3_downsample(x), ops_ref.median3x3_downsample(x))
        # NaN / inf / signed zeros in the windows: both sides order NaN behind +inf ("NaN sorts last"), bit-exact
        x.reshape(-1)[::7] = np.nan
        x.reshape(-1)[::11] = np.inf
        x.reshape(-1)[::13] = -np.inf
        x.reshape(-1)[::17] = -0.0
        x.reshape(-1)[3::5] = np.nan
        got, want = gpu_ctx.median3x3_downsample(x), ops_ref.median3x3_downsample(x)
        assert np.isnan(want).any() and np.isfinite(want).any()
        np.testing.assert_array_equal(got, want)      # equal_nan comparison; -0.0 == 0.0
    # evaluation.py:173: two median downsamples 192x256 -> 48x64
    x = rng.random((1, 3, 192, 256)).astype(np.float32)
    assert gpu_ctx.median3x3_downsample(gpu_ctx.median3x3_downsample(x)).shape == (1, 3, 48, 64)


def test_photometric_warp_kat_hip(gpu_ctx):
    """the reference-held photometric check of tests/test_pins.py on the HIP warp2d: image 2 of the sculpture pair pulled back
    by the reference flow resembles image 1 on the reference's visible mask; wrong sign / channel order / scale fail"""
    import test_pins
    g, img1, img2 = test_pins.golden()
    s = test_pins.photometric_scores(gpu_ctx.warp2d, g, img1, img2)
    test_pins.check_photometric(s)
    o = test_pins.photometric_scores(ops_ref.warp2d, g, img1, img2)
    for k in s:
        assert abs(s[k] - o[k]) < 1e-5, k


@pytest.mark.parametrize("inverse", [False, True])
def test_depth_to_normals(gpu_ctx, inverse):
    """HIP depth_to_normals (v2/losses.py:336-337) == oracle, incl. NaN pattern (border, invalid depth and its neighbours)"""
    import test_pins
    rng = np.random.default_rng(15)
    for (n, h, w) in ((2, 48, 64), (1, 192, 256), (3, 7, 9), (1, 3, 3), (1, 2, 5)):
        z = (1.0 + rng.random((n, 1, h, w))).astype(np.float32)
        z.reshape(-1)[::23] = 0
        z.reshape(-1)[5::41] = np.nan
        z.reshape(-1)[7::53] = -1
        z.reshape(-1)[9::59] = np.inf
        intr = np.tile(K_DEMON, (n, 1)) * (1 + 0.1 * rng.random((n, 4))).astype(np.float32)
        want = ops_ref.depth_to_normals(z, intr, inverse)
        got = gpu_ctx.depth_to_normals(z, intr, inverse)
        assert got.shape == (n, 3, h, w)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        m = ~np.isnan(want)
        if m.any():
            assert np.abs(got[m] - want[m]).max() < 2e-4
    nrm = np.array([0.3, -0.2, -0.9]) / np.linalg.norm([0.3, -0.2, -0.9])
    z = test_pins.plane_depth(nrm, nrm[2] * 2.0, 48, 64)
    got = gpu_ctx.depth_to_normals((1 / z if inverse else z)[None, None], K_DEMON, inverse)[0, :, 1:-1, 1:-1]
    np.testing.assert_allclose(got, np.broadcast_to(nrm[:, None, None], got.shape), atol=3e-4)


def test_ops_only_context():
    """demon_create_ops: the op entry points need no network context (stream + workspace only); network calls are refused"""
    from demon_amd import DemonContext, DemonError
    ctx = DemonContext.ops_only(0)
    try:
        rng = np.random.default_rng(16)
        x = rng.standard_normal((2, 3, 20, 30)).astype(np.float32)
        np.testing.assert_array_equal(ctx.leaky_relu(x, 0.1), ops_ref.leaky_relu(x, 0.1))
        np.testing.assert_array_equal(ctx.median3x3_downsample(x), ops_ref.median3x3_downsample(x))
        w = rng.standard_normal((3, 3, 3, 8)).astype(np.float32)
        b = rng.standard_normal(8).astype(np.float32)
        assert rel_l1(ctx.conv2d(x, w, b, (1, 1), True), ops_ref.conv2d_hwio(x, w, b, (1, 1), (1, 1), True)) < 1e-5
        assert ctx.variables() == []
        with pytest.raises(DemonError):
            ctx.run_full(1, 1)
    finally:
        ctx.close()


def test_sops_module_mirror(gpu_ctx, synth_weights):
    """the lmbspecialops-style module API (keyword names of the reference call sites)"""
    import demon_amd
    from demon_amd import sops
    rng = np.random.default_rng(14)
    inv_depth = (0.2 + rng.random((1, 1, 48, 64))).astype(np.float32)
    rot, tr = _motion(rng, 1)
    f = sops.depth_to_flow(intrinsics=K_DEMON[None], depth=inv_depth, rotation=rot, translation=tr, inverse_depth=True,
                           normalize_flow=True)
    assert rel_l1(f, ops_ref.depth_to_flow(inv_depth, K_DEMON, rot, tr, True, True)) < 1e-4
    d = sops.flow_to_depth(flow=f, intrinsics=K_DEMON[None], rotation=rot, translation=tr, normalized_flow=True, inverse_depth=True)
    assert rel_l1(d, inv_depth) < 2e-3
    d2 = sops.flow_to_depth2(flow=f, intrinsics=K_DEMON[None], rotation=rot, translation=tr, normalized_flow=True, inverse_depth=True)
    assert rel_l1(d2, inv_depth) < 2e-3
    img = rng.random((1, 3, 48, 64)).astype(np.float32)
    wz = sops.warp2d(input=img, displacements=f, normalized=True, border_mode="value")
    assert rel_l1(wz, ops_ref.warp2d(img, f, True, "value")) < 1e-4
    np.testing.assert_array_equal(sops.leaky_relu(img - 0.5, leak=0.1), ops_ref.leaky_relu(img - 0.5, 0.1))
    with pytest.raises(ValueError):
        sops.warp2d(img, f, border_mode="mirror")
    nrm = sops.depth_to_normals(inv_depth, K_DEMON[None], inverse_depth=True)
    want = ops_ref.depth_to_normals(inv_depth, K_DEMON, True)
    assert np.array_equal(np.isnan(nrm), np.isnan(want)) and np.nanmax(np.abs(nrm - want)) < 2e-4
    assert sops._ctx().max_batch == 0      # the module runs on an op-only context (no network arena behind an elementwise op)


@pytest.mark.parametrize("shape", [(2, 2, 48, 64), (3, 1, 17, 33), (1, 3, 192, 256)])
def test_pointwise_l2_loss(gpu_ctx, shape):
    """HIP pointwise_l2_loss (v2/losses.py:33-54) == the numpy restatement, with NaN / inf pixels in prediction and ground truth"""
    rng = np.random.default_rng(50)
    inp = rng.standard_normal(shape).astype(np.float32)
    gt = rng.standard_normal(shape).astype(np.float32)
    gt.reshape(-1)[::13] = np.nan
    inp.reshape(-1)[::101] = np.inf
    for eps in (0.0, 1e-3):
        got = gpu_ctx.pointwise_l2_loss(inp, gt, eps)
        want = ops_ref.pointwise_l2_loss(inp, gt, eps)
        assert abs(got - want) <= 1e-5 * abs(want), (got, want)


def test_ground_truth_preparation_mirror(gpu_ctx):
    """demon_amd.losses (forward values of v2/losses.py:312-356 prepare_ground_truth_tensors, :57-104 SIG + loss) on the HIP ops
    == the same composition of oracle ops; also covers the reference's POSITIONAL depth_to_flow call (depth first, :332)"""
    from demon_amd import losses, sops
    rng = np.random.default_rng(60)
    n, h, w = 2, 192, 256
    depth = (0.3 + rng.random((n, 1, h, w))).astype(np.float32)        # inverse depth
    depth.reshape(-1)[::97] = np.nan                                     # holes in the ground truth
    rot, tr = _motion(rng, n)
    intr = np.tile(K_DEMON, (n, 1))
    gt = losses.prepare_ground_truth_tensors(depth, rot, tr, intr)
    assert sorted(gt) == sorted(["depth0", "depth0_sig", "depth2", "depth2_sig", "flow0", "flow2", "flow2_sig", "flow5", "normal0", "normal2"])
    pyr = [depth]
    for _ in range(5):
        pyr.append(ops_ref.median3x3_downsample(pyr[-1]))
    np.testing.assert_array_equal(gt["depth2"], pyr[2])
    assert gt["depth2"].shape == (n, 1, 48, 64) and gt["flow5"].shape == (n, 2, 6, 8)

    def close(a, b, tol):
        assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b))
        m = ~np.isnan(b)
        assert np.abs(a[m] - b[m]).max() < tol

    for key, lvl in (("flow0", 0), ("flow2", 2), ("flow5", 5)):
        close(gt[key], ops_ref.depth_to_flow(pyr[lvl], K_DEMON, rot, tr, True, True), 1e-5)
    close(gt["normal0"], ops_ref.depth_to_normals(pyr[0], K_DEMON, True), 3e-4)
    close(gt["normal2"], ops_ref.depth_to_normals(pyr[2], K_DEMON, True), 3e-4)
    deltas = [1, 2, 4, 8, 16]
    want = np.concatenate([ops_ref.scale_invariant_gradient(pyr[2], [d], [1.0], 0.001) for d in deltas], axis=1)
    assert gt["depth2_sig"].shape == (n, 10, 48, 64)
    close(gt["depth2_sig"], want, 1e-5)
    # flow (C = 2): the op folds the two channels into the batch -> [2N, 5 deltas x (x, y), H, W], as the reference's tensors
    assert gt["flow2_sig"].shape == (2 * n, 10, 48, 64)
    fl2 = ops_ref.depth_to_flow(pyr[2], K_DEMON, rot, tr, True, True)
    want_f = np.concatenate([ops_ref.scale_invariant_gradient(fl2, [d], [1.0], 0.001) for d in deltas], axis=1)
    close(gt["flow2_sig"], want_f, 2e-4)
    # loss_flow2_sig of v2/losses.py:176-177: ONE pointwise l2 over the 10 channels, mean over 2N * H * W
    pr_f = gt["flow2_sig"] + rng.standard_normal(gt["flow2_sig"].shape).astype(np.float32) * 0.05
    pr_f[np.isnan(gt["flow2_sig"])] = 0.0
    got_f = losses.pointwise_l2_loss(pr_f, gt["flow2_sig"], 1e-3)
    ref_f = ops_ref.pointwise_l2_loss(pr_f, want_f, 1e-3)
    assert abs(got_f - ref_f) < 1e-4 * ref_f
    pred = gt["depth2_sig"] + rng.standard_normal(gt["depth2_sig"].shape).astype(np.float32) * 0.1
    got = losses.scale_invariant_gradient_loss(pred, gt["depth2_sig"], 1e-3)
    ref = sum(ops_ref.pointwise_l2_loss(pred[:, 2 * i:2 * i + 2], want[:, 2 * i:2 * i + 2], 1e-3) for i in range(5))
    assert abs(got - ref) < 1e-4 * ref
    f = sops.depth_to_flow(depth, intr, rot, tr, inverse_depth=True, normalize_flow=True)   # positional: depth first
    close(f, gt["flow0"], 1e-7)
