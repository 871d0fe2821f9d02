"""GPU parity of the v2 model (python/depthmotionnet/v2/networks.py over v2/blocks.py: padding='same', asymmetric separable
pairs, dense5 bottleneck, motion_conv3..5b branch, flow_to_depth2 + clip, predict_normal0) against the CPU oracle's v2
restatement, through the C ABI (demon_create_v2).  Gate: relative L1 <= 1e-3 per output tensor, as for the original model."""
import numpy as np
import pytest

from conftest import rel_l1, make_inputs
from oracle import net_ref, ops_ref

pytestmark = pytest.mark.gpu
TOL = 1e-3
KEYS = ("predict_flow5", "predict_flow2", "predict_depth2", "predict_normal2", "predict_rotation", "predict_translation")


@pytest.fixture(scope="module")
def v2_weights():
    from demon_amd import weights
    return weights.synthetic_weights(seed=1, version=2)


@pytest.fixture(scope="module")
def v2_ctx(v2_weights):
    from demon_amd import DemonContext
    ctx = DemonContext(device=0, max_batch=3, height=192, width=256, version=2)
    ctx.set_weights(v2_weights)
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def ref(v2_weights):
    return net_ref.DemonRefV2(v2_weights)


def _cmp(got, want, keys, tol=TOL):
    for k in keys:
        assert got[k].shape == want[k].shape, k
        assert np.isfinite(got[k]).all(), k
        err = rel_l1(got[k], want[k])
        assert err < tol, "%s rel L1 %.3e" % (k, err)


def test_v2_variable_table_matches_library(v2_ctx):
    from demon_amd import weights
    assert dict(v2_ctx.variables()) == weights.variable_shapes(version=2)
    assert v2_ctx.lib.demon_variant(v2_ctx.h) == 2


@pytest.mark.parametrize("cfg", [(6, 24, 9, 1, 2, 1), (24, 32, 1, 9, 1, 2), (32, 48, 7, 1, 2, 1), (64, 96, 5, 1, 2, 1),
                                 (16, 16, 1, 3, 1, 2), (8, 16, 3, 3, 2, 2), (8, 4, 3, 3, 1, 1)])
def test_v2_same_padding_conv_layer(v2_ctx, cfg):
    """demon_op_conv2d with padding='same' == the naive loops on an explicitly padded input (v2/helpers.py:24-91)"""
    cin, cout, kh, kw, sh, sw = cfg
    rng = np.random.default_rng(kh * 10 + kw)
    x = rng.standard_normal((2, cin, 24, 32)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    got = v2_ctx.conv2d(x, w, b, (sh, sw), lrelu=True, padding="same")
    want = ops_ref.conv2d_hwio_same(x, w, b, (sh, sw), True)
    assert got.shape == want.shape
    assert rel_l1(got, want) < 1e-5


@pytest.mark.parametrize("n", [1, 3])
def test_v2_bootstrap(v2_ctx, ref, n):
    pair, img2_2 = make_inputs(n, seed=20 + n)
    _cmp(v2_ctx.bootstrap(pair, img2_2), ref.bootstrap(pair, img2_2), KEYS + ("predict_conf5", "predict_conf2", "predict_scale"))


def test_v2_iterative(v2_ctx, ref):
    pair, img2_2 = make_inputs(2, seed=25)
    b = ref.bootstrap(pair, img2_2)
    args = (pair, img2_2, b["predict_depth2"], b["predict_normal2"], b["predict_rotation"], b["predict_translation"])
    _cmp(v2_ctx.iterative(*args), ref.iterative(*args), KEYS)


def test_v2_iterative_clip_and_gate(v2_ctx, ref):
    """flow_to_depth2 results outside [0, 50] are clipped (v2/blocks.py:379); bad motion is gated (v2/blocks.py:163-168)"""
    pair, img2_2 = make_inputs(2, seed=26)
    depth2 = np.full((2, 1, 48, 64), 0.5, np.float32)
    depth2[0, 0, :10] = -1.0
    normal2 = np.zeros((2, 3, 48, 64), np.float32)
    rot = np.array([[0.0, 0.0, 0.0], [0.3, -0.2, 0.1]], np.float32)
    tr = np.array([[1e-4, 0.0, 0.0], [0.1, 0.9, -0.2]], np.float32)   # tiny baseline -> huge inverse depth from flow -> clip
    args = (pair, img2_2, depth2, normal2, rot, tr)
    _cmp(v2_ctx.iterative(*args), ref.iterative(*args), KEYS)


def test_v2_clip_stress_weights(v2_ctx, v2_weights):
    """flows that disagree with the motion: the triangulated depth crosses zero at many pixels and clip(1/z, 0, 50) jumps between
    0 and 50 there (v2/blocks.py:362-381).  One stage from identical inputs still agrees in aggregate (cf. the gate stress test
    of the original model); chains of stages are chaotic in the model itself (demon_amd/weights.py synthetic_weights)."""
    from demon_amd import weights
    w = weights.synthetic_weights(seed=1, version=2, consistent_flow=False)
    v2_ctx.set_weights(w)
    try:
        ref = net_ref.DemonRefV2(w)
        pair, img2_2 = make_inputs(2, seed=25)
        b = ref.bootstrap(pair, img2_2)
        args = (pair, img2_2, b["predict_depth2"], b["predict_normal2"], b["predict_rotation"], b["predict_translation"])
        _cmp(v2_ctx.iterative(*args), ref.iterative(*args), KEYS, tol=5e-3)
    finally:
        v2_ctx.set_weights(v2_weights)


def test_v2_refine_depth_and_normals(v2_ctx, ref):
    pair, _ = make_inputs(2, seed=27)
    rng = np.random.default_rng(28)
    depth2 = (0.2 + rng.random((2, 1, 48, 64))).astype(np.float32)
    image1 = np.ascontiguousarray(pair[:, :3])
    _cmp(v2_ctx.refine(image1, depth2), ref.refine(image1, depth2), ("predict_depth0", "predict_normal0"))


def test_v2_full_pipeline(v2_ctx, ref):
    """bootstrap + 3 x iterative + refine (example_v2.py:93-105), device resident, graph == eager == staged"""
    pair, img2_2 = make_inputs(2, seed=29)
    want = ref.full(pair, img2_2, iterations=3)
    got = v2_ctx.full(pair, img2_2, iterations=3)
    keys = KEYS + ("predict_depth0", "predict_normal0")
    _cmp(got, want, keys)
    v2_ctx.set_option("hipgraph", 0)
    try:
        eager = v2_ctx.full(pair, img2_2, iterations=3)
    finally:
        v2_ctx.set_option("hipgraph", 1)
    v2_ctx.set_option("reuse_image_features", 1)
    try:
        reuse = v2_ctx.full(pair, img2_2, iterations=3)
    finally:
        v2_ctx.set_option("reuse_image_features", 0)
    for k in keys:
        np.testing.assert_array_equal(got[k], eager[k])
        np.testing.assert_array_equal(got[k], reuse[k])
    # staged host API (5 calls like example_v2.py) == device-resident loop, bit for bit
    r = v2_ctx.bootstrap(pair, img2_2)
    for _ in range(3):
        r = v2_ctx.iterative(pair, img2_2, r["predict_depth2"], r["predict_normal2"], r["predict_rotation"], r["predict_translation"])
    staged = v2_ctx.refine(np.ascontiguousarray(pair[:, :3]), r["predict_depth2"])
    np.testing.assert_array_equal(staged["predict_depth0"], got["predict_depth0"])
    np.testing.assert_array_equal(staged["predict_normal0"], got["predict_normal0"])


def test_v2_reference_api_mirror(v2_weights, ref):
    """depthmotionnet.v2.networks drop-in: constructor(session), eval signatures, keys and shapes (example_v2.py:81-105)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "python"))
    import demon_amd
    from depthmotionnet.v2.networks import BootstrapNet, IterativeNet, RefinementNet
    demon_amd.set_default_weights(v2_weights)
    pair, img2_2 = make_inputs(1, seed=30)
    boot, it, rf = BootstrapNet(None), IterativeNet(None), RefinementNet(None)
    r = boot.eval(pair, img2_2)
    assert sorted(r) == sorted(KEYS)
    for _ in range(3):
        r = it.eval(pair, img2_2, r["predict_depth2"], r["predict_normal2"], r["predict_rotation"], r["predict_translation"])
    out = rf.eval(pair[:, :3], r["predict_depth2"], r["predict_normal2"])
    assert sorted(out) == ["predict_depth0", "predict_normal0"]
    want = ref.full(pair, img2_2, iterations=3)
    _cmp(out, want, ("predict_depth0", "predict_normal0"))
    with pytest.raises(ValueError):
        rf.eval(pair[:, :3], r["predict_depth2"][:, :, :-1], r["predict_normal2"])


def test_normal0_is_v2_only(gpu_ctx):
    from demon_amd import DemonError
    with pytest.raises(DemonError):
        gpu_ctx._check(gpu_ctx.lib.demon_download_normal0(gpu_ctx.h, 1, np.empty((1, 3, 192, 256), np.float32).ctypes.data_as(
            __import__("ctypes").POINTER(__import__("ctypes").c_float))))
