"""GPU parity of the MFMA implicit-GEMM convolution family against the oracle, layer by layer, through
the C ABI (demon_op_conv2d / demon_op_deconv4x4s2 / demon_op_dense)."""
import numpy as np
import pytest

from conftest import rel_l1
from oracle import ops_ref

pytestmark = pytest.mark.gpu


def _torch_conv(x, w, b, stride, lrelu):
    import torch
    import torch.nn.functional as F
    kh, kw = w.shape[0], w.shape[1]
    xt = F.pad(torch.from_numpy(x), (kw // 2, kw // 2, kh // 2, kh // 2))
    y = F.conv2d(xt, torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), torch.from_numpy(b), stride=stride)
    if lrelu:
        y = torch.where(y >= 0, y, 0.1 * y)
    return y.numpy()


# (cin, cout, kh, kw, sh, sw, H, W): every distinct contraction shape of the five nets at 192x256
NET_CONVS = [
    (6, 32, 9, 1, 2, 1, 192, 256), (32, 32, 1, 9, 1, 2, 96, 256), (32, 64, 7, 1, 2, 1, 96, 128), (64, 64, 1, 7, 1, 2, 48, 128),
    (9, 32, 3, 1, 1, 1, 48, 64), (32, 32, 1, 3, 1, 1, 48, 64), (64, 64, 3, 1, 1, 1, 48, 64), (64, 128, 5, 1, 2, 1, 48, 64),
    (128, 128, 1, 5, 1, 2, 24, 64), (128, 128, 3, 1, 1, 1, 24, 32), (128, 256, 5, 1, 2, 1, 24, 32), (256, 256, 1, 5, 1, 2, 12, 32),
    (256, 256, 1, 3, 1, 1, 12, 16), (256, 512, 5, 1, 2, 1, 12, 16), (512, 512, 1, 5, 1, 2, 6, 16), (512, 512, 3, 1, 1, 1, 6, 8),
    (256, 512, 3, 1, 2, 1, 12, 16), (512, 24, 3, 3, 1, 1, 6, 8), (24, 4, 3, 3, 1, 1, 6, 8), (128, 24, 3, 3, 1, 1, 48, 64),
    (24, 4, 3, 3, 1, 1, 48, 64), (512, 128, 3, 3, 1, 1, 6, 8), (4, 32, 3, 3, 1, 1, 192, 256), (32, 64, 3, 3, 2, 2, 192, 256),
    (64, 64, 3, 3, 1, 1, 96, 128), (64, 128, 3, 3, 2, 2, 96, 128), (128, 128, 3, 3, 1, 1, 48, 64), (64, 16, 3, 3, 1, 1, 192, 256),
    (16, 1, 3, 3, 1, 1, 192, 256),
]


@pytest.mark.parametrize("cfg", NET_CONVS)
def test_conv_shapes_of_the_nets(gpu_ctx, cfg):
    cin, cout, kh, kw, sh, sw, H, W = cfg
    rng = np.random.default_rng(20)
    n = 2
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    got = gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=True)
    want = _torch_conv(x, w, b, (sh, sw), True)
    assert got.shape == want.shape
    assert rel_l1(got, want) < 1e-5


@pytest.mark.parametrize("cfg", [(3, 5, 3, 3, 1, 1, 7, 9), (5, 33, 3, 1, 2, 1, 9, 5), (7, 3, 1, 5, 1, 2, 4, 11), (1, 1, 9, 1, 2, 1, 5, 3)])
def test_conv_ragged_vs_naive_c(gpu_ctx, cfg):
    """odd sizes / channel counts not multiples of the tile, against the double-accumulating C loops"""
    cin, cout, kh, kw, sh, sw, H, W = cfg
    rng = np.random.default_rng(21)
    for n in (1, 3):
        x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
        w = rng.standard_normal((kh, kw, cin, cout)).astype(np.float32)
        b = rng.standard_normal((cout,)).astype(np.float32)
        for lrelu in (False, True):
            got = gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=lrelu)
            want = ops_ref.conv2d_hwio(x, w, b, (sh, sw), (kh // 2, kw // 2), lrelu)
            assert got.shape == want.shape
            assert rel_l1(got, want) < 1e-5


def test_conv_is_transpose_detecting(gpu_ctx):
    """A = identity-like weights with an asymmetric input: catches swapped rows/cols in the MFMA layouts"""
    cin = cout = 32
    x = np.arange(2 * cin * 4 * 40, dtype=np.float32).reshape(2, cin, 4, 40) * 1e-3
    w = np.zeros((1, 1, cin, cout), np.float32)
    for i in range(cin):
        w[0, 0, i, (i * 7 + 3) % cout] = 1.0 + i  # permutation with distinct gains
    b = np.arange(cout, dtype=np.float32)
    got = gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=False)
    want = ops_ref.conv2d_hwio(x, w, b, (1, 1), (0, 0), False)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("cfg", [(512, 256, 6, 8), (514, 128, 12, 16), (256, 64, 24, 32), (4, 2, 6, 8), (128, 64, 48, 64), (128, 32, 96, 128), (5, 3, 3, 5)])
def test_deconv(gpu_ctx, cfg):
    import torch
    import torch.nn.functional as F
    cin, cout, H, W = cfg
    rng = np.random.default_rng(22)
    n = 2
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((4, 4, cout, cin)) / np.sqrt(4 * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    got = gpu_ctx.deconv4x4s2(x, w, b, lrelu=True)
    y = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), torch.from_numpy(b), stride=2, padding=1)
    want = torch.where(y >= 0, y, 0.1 * y).numpy()
    assert got.shape == want.shape == (n, cout, 2 * H, 2 * W)
    assert rel_l1(got, want) < 1e-5
    if cin <= 8:
        assert rel_l1(got, ops_ref.deconv4x4s2_crop(x, w, b, True)) < 1e-5


@pytest.mark.parametrize("cfg", [(6144, 1024), (1024, 128), (128, 7), (37, 5)])
def test_dense(gpu_ctx, cfg):
    cin, cout = cfg
    rng = np.random.default_rng(23)
    for n in (1, 4):
        x = rng.standard_normal((n, cin)).astype(np.float32)
        w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
        b = rng.standard_normal((cout,)).astype(np.float32)
        got = gpu_ctx.dense(x, w, b, lrelu=True)
        want = ops_ref.dense(x, w, b, True)
        assert rel_l1(got, want) < 1e-5


def test_nan_propagates_like_the_reference(gpu_ctx):
    """0 * NaN = NaN inside a contraction (TF and the MFMA agree); zero padding stays zero"""
    x = np.ones((1, 2, 4, 4), np.float32)
    x[0, 0, 1, 1] = np.nan
    w = np.ones((3, 3, 2, 1), np.float32)
    b = np.zeros((1,), np.float32)
    got = gpu_ctx.conv2d(x, w, b, (1, 1))
    want = ops_ref.conv2d_hwio(x, w, b, (1, 1), (1, 1), False)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert got[0, 0, 3, 3] == want[0, 0, 3, 3] == 8.0
