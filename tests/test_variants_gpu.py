"""Every kernel variant the autotuner may pick (im2col / patch-staged, each tile shape, several split-K factors)
must give the same layer result.  DEMON_FORCE_PLAN is a test hook read by the library at launch time."""
import os

import numpy as np
import pytest

from conftest import rel_l1

pytestmark = pytest.mark.gpu

# (kind, cin, cout, kh, kw, sh, sw, H, W)
LAYERS = [
    ("conv", 32, 32, 1, 9, 1, 2, 12, 64), ("conv", 64, 64, 3, 1, 1, 1, 48, 64), ("conv", 64, 128, 5, 1, 2, 1, 48, 64),
    ("conv", 128, 128, 1, 5, 1, 2, 24, 64), ("conv", 256, 256, 3, 1, 1, 1, 12, 16), ("conv", 512, 512, 1, 3, 1, 1, 6, 8),
    ("conv", 256, 512, 5, 1, 2, 1, 12, 16), ("conv", 32, 64, 7, 1, 2, 1, 24, 32), ("conv", 128, 24, 3, 3, 1, 1, 48, 64),
    ("conv", 64, 64, 3, 3, 1, 1, 24, 32), ("conv", 32, 64, 3, 3, 2, 2, 24, 32), ("conv", 512, 24, 3, 3, 1, 1, 6, 8),
    ("conv", 6, 32, 9, 1, 2, 1, 48, 64), ("conv", 24, 4, 3, 3, 1, 1, 6, 8), ("conv", 24, 4, 3, 3, 1, 1, 48, 64),
    ("conv", 16, 1, 3, 3, 1, 1, 40, 136), ("conv", 5, 3, 3, 3, 1, 1, 9, 7), ("conv", 24, 2, 3, 3, 1, 1, 17, 130),
    ("conv", 4, 32, 3, 3, 1, 1, 40, 72), ("conv", 5, 32, 9, 1, 2, 1, 48, 64), ("conv", 3, 20, 3, 3, 1, 1, 17, 33), ("conv", 6, 32, 1, 9, 1, 2, 24, 128),
    ("conv", 64, 16, 3, 3, 1, 1, 48, 64), ("conv", 64, 16, 3, 3, 1, 1, 21, 37), ("conv", 30, 12, 3, 3, 2, 2, 24, 32),
    ("deconv", 512, 256, 0, 0, 0, 0, 6, 8), ("deconv", 514, 128, 0, 0, 0, 0, 12, 16), ("deconv", 128, 32, 0, 0, 0, 0, 24, 32),
    ("deconv", 128, 64, 0, 0, 0, 0, 17, 35), ("deconv", 4, 2, 0, 0, 0, 0, 6, 8), ("deconv", 30, 40, 0, 0, 0, 0, 9, 50),
    ("conv", 16, 40, 3, 3, 1, 1, 7, 9), ("conv", 48, 32, 1, 5, 1, 2, 5, 23), ("deconv", 16, 8, 0, 0, 0, 0, 3, 5),
]


def _ref(kind, x, w, b, stride):
    import torch
    import torch.nn.functional as F
    if kind == "deconv":
        y = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), torch.from_numpy(b), stride=2, padding=1)
    else:
        kh, kw = w.shape[0], w.shape[1]
        xt = F.pad(torch.from_numpy(x), (kw // 2, kw // 2, kh // 2, kh // 2))
        y = F.conv2d(xt, torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), torch.from_numpy(b), stride=stride)
    return torch.where(y >= 0, y, 0.1 * y).numpy()


@pytest.mark.parametrize("layer", LAYERS)
def test_all_variants_agree(gpu_ctx, layer):
    kind, cin, cout, kh, kw, sh, sw, H, W = layer
    rng = np.random.default_rng(30)
    n = 3
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    if kind == "deconv":
        w = (rng.standard_normal((4, 4, cout, cin)) / np.sqrt(4 * cin)).astype(np.float32)
    else:
        w = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref(kind, x, w, b, (sh, sw))
    plans = [(3, 0, 0)] + [(0, t, ks) for t in range(8) for ks in (1, 2, 3, 5)] + [(1, t, ks) for t in range(9) for ks in (0, 2, 3, 5)]
    plans += [(4, v, ks) for v in range(18) for ks in (1, 2, 3, 5)]   # register-streaming kernel (applies when Cin % 16 == 0)
    plans += [(5, v, ks) for v in range(22) for ks in (1, 2, 3, 5)]    # fragment-tiled kernel (same requirement)
    plans += [(8, v, ks) for v in range(7) for ks in (1, 2, 3)]   # minimal-filtering transposed conv (conv_wino.hip; Cin >= 16)
    plans += [(10, v, ks) for v in range(13) for ks in (1, 2)]     # 1-D minimal filtering (3 taps stride 1; 5 / 7 / 9 taps stride 2)
    try:
        for plan in plans:
            os.environ["DEMON_FORCE_PLAN"] = "%d,%d,%d" % plan
            got = gpu_ctx.deconv4x4s2(x, w, b, lrelu=True) if kind == "deconv" else gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=True)
            err = rel_l1(got, want)
            assert err < 1e-5, "plan %s: rel L1 %.3e" % (plan, err)
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


def test_streaming_kernel_is_bit_identical_to_im2col_and_runs_dense(gpu_ctx):
    """conv_stream.hip reduces in the same order as conv_mfma.hip (tap major, channel minor), so without split-K the two give
    the same bits; the dense layers (H = W = 1, one tap) run on it too"""
    rng = np.random.default_rng(31)
    x = rng.standard_normal((5, 256, 12, 16)).astype(np.float32)
    w = (rng.standard_normal((3, 1, 256, 256)) / 28).astype(np.float32)
    b = rng.standard_normal(256).astype(np.float32)
    try:
        os.environ["DEMON_FORCE_PLAN"] = "0,6,1"
        ref = gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True)
        for v in range(10):   # variants without in-workgroup split-K keep the im2col summation order
            os.environ["DEMON_FORCE_PLAN"] = "4,%d,1" % v
            np.testing.assert_array_equal(gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True), ref)
        for v in range(14):   # ... and so does the fragment-tiled kernel
            os.environ["DEMON_FORCE_PLAN"] = "5,%d,1" % v
            np.testing.assert_array_equal(gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True), ref)
        xd = rng.standard_normal((7, 6144)).astype(np.float32)
        wd = (rng.standard_normal((6144, 1024)) / 78).astype(np.float32)
        bd = rng.standard_normal(1024).astype(np.float32)
        want = xd.astype(np.float64) @ wd.astype(np.float64) + bd
        want = np.where(want >= 0, want, 0.1 * want)
        for plan in ("4,0,1", "4,0,8", "4,4,48", "4,2,16", "4,12,1", "4,11,2", "4,16,1"):
            os.environ["DEMON_FORCE_PLAN"] = plan
            assert rel_l1(gpu_ctx.dense(xd, wd, bd, lrelu=True), want) < 1e-5, plan
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


def test_split_k_is_always_finished_by_the_reduce_launch(gpu_ctx):
    """The "K slices combined inside the launch" form (ksplit + 1000 on kinds 0 / 4 / 5, rounds 2-3) is gone: its hand-off rested on
    write-through stores and a relaxed ticket, not on a release / acquire pair, and it was never faster than the reduce launch.  A
    plan that still asks for it is rejected, and split-K results are reproducible bit for bit (fixed slice order in the reduce)."""
    from demon_amd import DemonContext
    from demon_amd.engine import DemonError
    ctx = DemonContext(0, 2, 192, 256)
    try:
        name = "netFlow1/conv5_1y"
        for kind, tile in ((0, 6), (4, 0), (5, 0)):
            with pytest.raises(DemonError):
                ctx.set_plan(2, {name: [kind, tile, 1004]})
            ctx.set_plan(2, {name: [kind, tile, 4]})
    finally:
        ctx.close()
    rng = np.random.default_rng(33)
    x = rng.standard_normal((24, 512, 6, 8)).astype(np.float32)
    w = (rng.standard_normal((3, 1, 512, 512)) / 39).astype(np.float32)
    b = rng.standard_normal(512).astype(np.float32)
    try:
        for kind, tile, ks in ((5, 6, 4), (4, 9, 4), (0, 6, 4)):
            os.environ["DEMON_FORCE_PLAN"] = "%d,%d,%d" % (kind, tile, ks)
            ref = gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True)
            for rep in range(4):
                np.testing.assert_array_equal(gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True), ref)
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


WINO_LAYERS = [(512, 256, 6, 8), (514, 128, 12, 16), (258, 64, 24, 32), (128, 32, 24, 32), (128, 64, 17, 35), (30, 40, 9, 50), (16, 8, 3, 5),
               (130, 20, 5, 3), (64, 48, 30, 40)]


@pytest.mark.parametrize("shape", WINO_LAYERS)
def test_minimal_filtering_deconv(gpu_ctx, shape):
    """conv_wino.hip: F(2,2) x F(2,2) per sub-pixel class of the 4x4 stride-2 transposed conv (blocks_original.py:64-75) computes
    the same sums with 9 instead of 16 multiplications per 2 x 2 output block -- other summation order than the direct kernels,
    so compared with PyTorch to 1e-5 like every other variant, for every tile count and split-K, on network shapes, odd sizes,
    Cin not a multiple of 4 and Cout not a multiple of 16; the tag proves the forced variant really ran"""
    cin, cout, H, W = shape
    rng = np.random.default_rng(32)
    n = 5
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((4, 4, cout, cin)) / np.sqrt(4 * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("deconv", x, w, b, (2, 2))
    try:
        os.environ["DEMON_FORCE_PLAN"] = "1,8,0"
        direct = gpu_ctx.deconv4x4s2(x, w, b, lrelu=True)
        for v in range(7):   # 4, 5: two 16-channel blocks per wave (the packed weights are padded to a multiple of 32 channels); 6: K halves inside the workgroup
            for ks in (1, 2, 5):
                if v == 6 and ks > 1:
                    continue
                os.environ["DEMON_FORCE_PLAN"] = "8,%d,%d" % (v, ks)
                got = gpu_ctx.deconv4x4s2(x, w, b, lrelu=True)
                tag = gpu_ctx.last_kernel()
                if v == 6 and not tag.startswith("wino_deconv<16x48,kh2>"):
                    assert (cin % 8 or W % 2) and shape != WINO_LAYERS[0], (shape, tag)   # (needs two equal halves of whole K-steps and an even width; refine4's shape must run it)
                elif v >= 4 and shape not in WINO_LAYERS[:5] and not tag.startswith("wino_deconv<"):
                    pass   # (a second weight tile does not fit the 64 KB of LDS beside five small images' patches: another kernel ran; same checks)
                else:
                    assert tag.startswith("wino_deconv<%dx%d" % ((16, 32), (16, 64), (16, 48), (16, 16), (32, 16), (32, 32), (16, 48))[v]), tag
                    assert ("+splitk" in tag) == (ks > 1), tag
                err = rel_l1(got, want)
                assert err < 1e-5, "variant %d split %d: rel L1 %.3e" % (v, ks, err)
                assert rel_l1(got, direct) < 1e-5
                np.testing.assert_array_equal(got, gpu_ctx.deconv4x4s2(x, w, b, lrelu=True))   # deterministic
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


# (cin, cout, kh, kw, sh, sw, H, W)
WINO1D_LAYERS = [(64, 64, 3, 1, 1, 1, 24, 32), (128, 128, 1, 3, 1, 1, 12, 16), (64, 128, 5, 1, 2, 1, 48, 64), (128, 128, 1, 5, 1, 2, 24, 64),
                 (32, 64, 7, 1, 2, 1, 24, 32), (32, 32, 1, 7, 1, 2, 12, 64), (32, 32, 1, 9, 1, 2, 12, 64), (16, 32, 9, 1, 2, 1, 48, 64),
                 (18, 40, 3, 1, 1, 1, 7, 9), (30, 24, 1, 5, 1, 2, 5, 23), (20, 36, 7, 1, 2, 1, 17, 33), (256, 256, 3, 1, 1, 1, 12, 16), (34, 32, 1, 9, 1, 2, 9, 31), (34, 32, 1, 9, 1, 2, 9, 36), (30, 24, 1, 5, 1, 2, 5, 24), (32, 32, 1, 7, 1, 2, 6, 20),
                 (130, 24, 3, 3, 1, 1, 48, 64), (64, 16, 3, 3, 1, 1, 40, 72), (32, 16, 3, 1, 1, 1, 24, 32), (22, 8, 1, 3, 1, 1, 12, 16), (16, 12, 1, 5, 1, 2, 12, 32), (64, 64, 3, 3, 1, 1, 24, 32), (18, 40, 3, 3, 1, 1, 7, 9)]   # 3 x 3: three 1 x 3 filters


@pytest.mark.parametrize("layer", WINO1D_LAYERS)
def test_minimal_filtering_1d(gpu_ctx, layer):
    """conv_wino.hip / wino1d_tables.h: two outputs along the filter axis per window with k + 1 (stride 1, k = 3) or k + 2 (stride 2,
    polyphase F(2,re) + F(2,ro), k = 5 / 7 / 9) multiplications instead of 2k, for the k x 1 / 1 x k layers of convrelu2
    (helpers.py:105-153).  Other rounding than the direct kernels (integer-valued transforms, denominators up to 24 in the output
    transform; a CPU check of the generated tables in exact rationals is part of tools/gen_wino1d.py): the bar stays 1e-5 relative L1
    against PyTorch, every workgroup shape and split-K, odd sizes, Cin not a multiple of 4"""
    cin, cout, kh, kw, sh, sw, H, W = layer
    rng = np.random.default_rng(34)
    n = 5
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("conv", x, w, b, (sh, sw))
    ran = 0
    try:
        for v in range(13):
            for ks in (1, 2, 3):
                os.environ["DEMON_FORCE_PLAN"] = "10,%d,%d" % (v, ks)
                got = gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=True)
                tag = gpu_ctx.last_kernel()
                if not tag.startswith("wino1d<"):
                    continue   # shape not built for this filter (accumulator budget) or Cout not a multiple of its channel tile
                ran += 1
                err = rel_l1(got, want)
                assert err < 1e-5, "variant %d split %d (%s): rel L1 %.3e" % (v, ks, tag, err)
                np.testing.assert_array_equal(got, gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=True))   # deterministic
        if (kh == 1 or kh == 3 == kw) and W % (2 * sw):   # filters along x load 8- / 16-byte vectors: rows must be a multiple of 2 / 4 pixels, else the layer
            assert ran == 0, ran            # stays on the direct kernels (checked above all the same: the forced plan falls back)
        else:
            assert ran >= 3, ran
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


# (batch, cin, cout)
DENSE_LAYERS = [(32, 4608, 4608), (7, 6144, 1024), (1, 8192, 128), (33, 4096, 256), (64, 4608, 384), (5, 8200, 128)]


@pytest.mark.parametrize("layer", DENSE_LAYERS)
def test_weight_streaming_dense(gpu_ctx, layer):
    """dense_stream.hip (plan kind 11): dense5 of the v2 blocks (v2/blocks.py:197-213) and motion_fc1 (blocks_original.py:390-396) as
    a weight stream -- weights global -> registers -> MFMA, 4 waves x K slices combined in a fixed order.  Against float64 numpy,
    every split, batches that are not a multiple of the 32-sample MFMA block (and more than one block), K ranges that do not
    divide evenly over waves and slices; two runs are bit-identical"""
    n, cin, cout = layer
    rng = np.random.default_rng(35)
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    want = x.astype(np.float64) @ w.astype(np.float64) + b
    want = np.where(want >= 0, want, 0.1 * want)
    try:
        for ks in (1, 2, 3, 5, 9, 16, 36, 64):
            os.environ["DEMON_FORCE_PLAN"] = "11,%d,%d" % (ks & 1, ks)   # variant 1 = non-temporal weight loads
            got = gpu_ctx.dense(x, w, b, lrelu=True)
            tag = gpu_ctx.last_kernel()
            assert tag.startswith("dense_stream<"), tag
            err = rel_l1(got, want)
            assert err < 1e-5, "split %d (%s): rel L1 %.3e" % (ks, tag, err)
            np.testing.assert_array_equal(got, gpu_ctx.dense(x, w, b, lrelu=True))
        os.environ["DEMON_FORCE_PLAN"] = "11,0,1"
        got = gpu_ctx.dense(x, w, b, lrelu=False)   # no activation
        assert rel_l1(got, x.astype(np.float64) @ w.astype(np.float64) + b) < 1e-5
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


# (cin, cout, H, W)
THIN_LAYERS = [(6, 32, 192, 256), (6, 24, 96, 128), (6, 32, 50, 68), (3, 32, 34, 64), (5, 17, 18, 200)]


@pytest.mark.parametrize("layer", THIN_LAYERS)
def test_first_layer_weights_in_registers(gpu_ctx, layer):
    """conv_thin.hip (plan kind 12): the 9 x 1 stride-(2,1) conv over the <= 6 channels of image_pair (blocks_original.py:141, :331 via
    helpers.py:105-153) with the whole reduction in registers.  Against PyTorch; image borders, partial row / column tiles, fewer
    input / output channels than the tile holds"""
    cin, cout, H, W = layer
    rng = np.random.default_rng(36)
    n = 3
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((9, 1, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("conv", x, w, b, (2, 1))
    try:
        os.environ["DEMON_FORCE_PLAN"] = "12,0,1"
        got = gpu_ctx.conv2d(x, w, b, (2, 1), lrelu=True)
        assert gpu_ctx.last_kernel().startswith("conv_thin<"), gpu_ctx.last_kernel()
        assert got.shape == want.shape
        assert rel_l1(got, want) < 1e-5
        np.testing.assert_array_equal(got, gpu_ctx.conv2d(x, w, b, (2, 1), lrelu=True))
        lin = gpu_ctx.conv2d(x, w, b, (2, 1), lrelu=False)   # without the activation: the same sums
        np.testing.assert_array_equal(np.where(lin >= 0, lin, np.float32(0.1) * lin), got)
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


# (cin, cout, taps, H, W)
ROW_LAYERS = [(32, 32, 9, 96, 256), (32, 32, 7, 48, 128), (24, 32, 9, 20, 136), (32, 24, 7, 7, 64), (30, 32, 9, 5, 72), (16, 16, 9, 6, 260)]


@pytest.mark.parametrize("layer", ROW_LAYERS)
def test_row_conv_out_of_lds(gpu_ctx, layer):
    """conv_row.hip (plan kind 13): the 1 x 9 / 1 x 7 stride-(1,2) convs with <= 32 channels on both sides (conv1x of every block,
    conv2x of the iterative nets; helpers.py:105-153) in the minimal-filtering form of wino1d_tables.h with the whole reduction out
    of LDS.  Against PyTorch: image borders, partial row / column tiles, channel counts below the tile's and not a multiple of 4"""
    cin, cout, taps, H, W = layer
    rng = np.random.default_rng(37)
    n = 3
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((1, taps, cin, cout)) / np.sqrt(taps * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("conv", x, w, b, (1, 2))
    try:
        os.environ["DEMON_FORCE_PLAN"] = "13,0,1"
        got = gpu_ctx.conv2d(x, w, b, (1, 2), lrelu=True)
        assert gpu_ctx.last_kernel().startswith("conv_row<"), gpu_ctx.last_kernel()
        assert got.shape == want.shape
        assert rel_l1(got, want) < 1e-5
        np.testing.assert_array_equal(got, gpu_ctx.conv2d(x, w, b, (1, 2), lrelu=True))
        lin = gpu_ctx.conv2d(x, w, b, (1, 2), lrelu=False)
        np.testing.assert_array_equal(np.where(lin >= 0, lin, np.float32(0.1) * lin), got)
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


# (cin, cout, H, W)
WINO3_LAYERS = [(128, 24, 48, 64), (64, 16, 40, 72), (64, 64, 24, 64), (130, 24, 17, 34), (18, 40, 7, 66), (32, 16, 192, 256), (20, 64, 13, 128), (16, 8, 6, 64), (64, 64, 9, 44)]


@pytest.mark.parametrize("layer", WINO3_LAYERS)
def test_minimal_filtering_3x3_rows_stationary(gpu_ctx, layer):
    """conv_wino3.hip: 3 x 3 stride-1 convs as three 1 x 3 row filters with the transformed input rows kept in LDS for the three
    output rows they feed (plan kind 15): variants 0 .. 7 on F(2,3) tiles of two pixels (same products as the wino1d kernel's
    three-pass form, other order of addition), variants 8 .. 15 on F(4,3) tiles of four pixels (6 products per 4 outputs and kernel
    row; interpolation points 0, +-1, +-2: about twice the rounding error of a direct fp32 sum).  1e-5 relative L1 against PyTorch for
    every workgroup shape that fits the layer; ragged heights / widths, Cin not a multiple of 4 / 8, Cout below a channel block;
    deterministic."""
    cin, cout, H, W = layer
    rng = np.random.default_rng(35)
    n = 3
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("conv", x, w, b, (1, 1))
    ran = 0
    try:
        forms = set()
        for v in range(16):
            os.environ["DEMON_FORCE_PLAN"] = "15,%d,1" % v
            got = gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True)
            tag = gpu_ctx.last_kernel()
            if not tag.startswith("wino3rows<"):
                continue   # the shape does not fit this layer (channel block vs Cout, too many empty tile slots)
            ran += 1
            forms.add(tag.split(",")[0])
            err = rel_l1(got, want)
            assert err < 1e-5, "variant %d (%s): rel L1 %.3e" % (v, tag, err)
            np.testing.assert_array_equal(got, gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True))
        assert ran >= 1, layer
        if W % 4 == 0 and (cout > 16 or W >= 128):   # (the one-channel-block F(4,3) shapes are 256 pixels wide)
            assert forms == {"wino3rows<t3x3", "wino3rows<f4t3x3"}, forms   # both forms ran
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


# (cin, cout, H, W)
WINO3S2_LAYERS = [(32, 64, 48, 128), (64, 128, 24, 64), (20, 40, 13, 64), (18, 64, 9, 72), (64, 128, 17, 136), (16, 64, 8, 64), (32, 64, 96, 256)]


@pytest.mark.parametrize("layer", WINO3S2_LAYERS)
def test_minimal_filtering_3x3_stride2_rows(gpu_ctx, layer):
    """conv_wino3.hip, stride-2 form (plan kind 15, variants 16 .. 19; the refinement net's conv1 / conv2, blocks_original.py:484-511):
    output row r from the transformed input rows 2r - 1, 2r, 2r + 1 kept in LDS, along x the polyphase split with F(4,2) on the even
    and F(4,1) on the odd samples of a 9-pixel window -- 9 products per 4 outputs and kernel row instead of 12.  1e-5 relative L1
    against PyTorch for every workgroup shape that fits; odd heights (the zero row below the image), Cin not a multiple of 4, Cout
    below / across channel blocks, widths that leave tile columns empty; deterministic; the stride-1 variants refuse such a layer."""
    cin, cout, H, W = layer
    rng = np.random.default_rng(36)
    n = 3
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("conv", x, w, b, (2, 2))
    ran = []
    try:
        for v in range(20):
            os.environ["DEMON_FORCE_PLAN"] = "15,%d,1" % v
            got = gpu_ctx.conv2d(x, w, b, (2, 2), lrelu=True)
            tag = gpu_ctx.last_kernel()
            if v < 16:
                assert not tag.startswith("wino3rows<"), tag      # stride-1 forms do not take a stride-2 layer
            if not tag.startswith("wino3rows<"):
                continue
            assert tag == "wino3rows<s2t3x3,v%d>" % v, tag
            ran.append(v)
            err = rel_l1(got, want)
            assert err < 1e-5, "variant %d (%s): rel L1 %.3e" % (v, tag, err)
            np.testing.assert_array_equal(got, gpu_ctx.conv2d(x, w, b, (2, 2), lrelu=True))
            lin = gpu_ctx.conv2d(x, w, b, (2, 2), lrelu=False)
            np.testing.assert_array_equal(np.where(lin >= 0, lin, np.float32(0.1) * lin), got)
        assert ran, layer
        if (cin, cout) == (32, 64):
            assert set(ran) >= {16, 17, 18}, ran
        if (cin, cout) == (64, 128) and W == 64:
            assert set(ran) >= {16, 19}, ran
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


# (cin, cout, kh, kw, sh, sw, H, W)
WINO4_LAYERS = [(64, 64, 3, 1, 1, 1, 48, 64), (64, 64, 1, 3, 1, 1, 48, 64), (128, 128, 3, 1, 1, 1, 24, 32), (128, 128, 1, 3, 1, 1, 24, 32),
                (64, 128, 5, 1, 2, 1, 48, 64), (128, 128, 1, 5, 1, 2, 24, 64), (128, 256, 5, 1, 2, 1, 24, 32), (256, 256, 1, 5, 1, 2, 12, 32),
                (18, 40, 3, 1, 1, 1, 13, 70), (30, 24, 1, 5, 1, 2, 6, 256), (22, 36, 5, 1, 2, 1, 35, 66), (20, 16, 1, 3, 1, 1, 7, 64), (256, 256, 3, 1, 1, 1, 12, 16),
                (256, 256, 1, 3, 1, 1, 12, 16), (24, 48, 1, 5, 1, 2, 9, 32),
                (128, 128, 3, 1, 1, 1, 24, 32), (64, 64, 3, 1, 1, 1, 20, 64), (36, 64, 5, 1, 2, 1, 46, 32)]


@pytest.mark.parametrize("layer", WINO4_LAYERS)
def test_minimal_filtering_four_outputs_per_window(gpu_ctx, layer):
    """conv_wino4.hip (plan kind 16): k x 1 / 1 x k layers with four outputs per window -- F(4,3) for 3 taps stride 1 (6 products
    instead of 12), polyphase F(4,3) + F(4,2) for 5 taps stride 2 (11 instead of 20); interpolation points 0, +-1, +-2 / 0, +-1, 2
    (tables checked in exact rationals by tools/gen_wino1d.py).  1e-5 relative L1 against PyTorch for every workgroup shape that fits;
    ragged sizes, Cin not a multiple of 4 / 8 / 16, Cout not a multiple of the channel block; deterministic."""
    cin, cout, kh, kw, sh, sw, H, W = layer
    rng = np.random.default_rng(36)
    n = 3
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("conv", x, w, b, (sh, sw))
    ran = walked = 0
    try:
        for v in range(14):     # 9 .. 13: the three-lines-per-wave shapes of round 6 (staging units that do not divide the threads)
            plain = None
            for mode in (1, 2, 3):   # 2: tile-walking workgroups (round 6; falls back to the plain launch where that form does not exist / one round suffices); 3: flat line order (falls back where it saves no line block)
                os.environ["DEMON_FORCE_PLAN"] = "16,%d,%d" % (v, mode)
                got = gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=True)
                tag = gpu_ctx.last_kernel()
                if not tag.startswith("wino4<"):
                    break   # shape not built for this filter (accumulator budget), Cout not a multiple of its channel block, too much waste
                ran += 1
                err = rel_l1(got, want)
                assert err < 1e-5, "variant %d (%s): rel L1 %.3e" % (v, tag, err)
                np.testing.assert_array_equal(got, gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=True))
                if mode == 1:
                    plain = got
                else:   # (another template instance: the compiler contracts its epilogue's multiply-adds its own way -- last-bit differences)
                    walked += ",walk" in tag
                    assert rel_l1(got, plain) < 1e-6, tag
        assert ran >= 1, layer
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


# (cin, cout, kh, kw, sh, sw, H, W, batch)
WINO4_FLAT_LAYERS = [(256, 256, 1, 3, 1, 1, 12, 16, 7), (256, 256, 3, 1, 1, 1, 12, 16, 7), (256, 256, 1, 5, 1, 2, 12, 32, 5), (128, 256, 5, 1, 2, 1, 24, 32, 5),
                     (512, 512, 1, 3, 1, 1, 6, 8, 9), (512, 512, 1, 5, 1, 2, 6, 16, 9), (64, 48, 3, 1, 1, 1, 6, 8, 11), (40, 24, 1, 3, 1, 1, 5, 8, 6),
                     (36, 64, 5, 1, 2, 1, 10, 16, 7), (20, 40, 1, 5, 1, 2, 3, 32, 13), (128, 128, 1, 3, 1, 1, 24, 32, 3), (24, 32, 3, 1, 1, 1, 9, 16, 10)]


@pytest.mark.parametrize("layer", WINO4_FLAT_LAYERS)
def test_flat_line_order_equals_the_per_image_tiles(gpu_ctx, layer):
    """conv_wino4.hip, plan field ksplit = 3 (round 6): the lines of all images of the batch form one sequence, so that maps whose lines per
    image do not fill a workgroup's lines (12 x 16: 12 rows or 3 tile rows against 8 / 16) leave no empty slots -- a tile then holds lines of
    several images, every staging unit and every store carries its image in the buffer offset.  Batches that end in a ragged block, odd map
    sizes, Cin / Cout that do not fill the channel blocks; against PyTorch, against the per-image launch, deterministic."""
    cin, cout, kh, kw, sh, sw, H, W, n = layer
    rng = np.random.default_rng(77)
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("conv", x, w, b, (sh, sw))
    flat = 0
    try:
        for v in range(14):
            os.environ["DEMON_FORCE_PLAN"] = "16,%d,3" % v
            got = gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=True)
            tag = gpu_ctx.last_kernel()
            if ",flat>" not in tag:
                continue
            flat += 1
            err = rel_l1(got, want)
            assert err < 1e-5, "variant %d (%s): rel L1 %.3e" % (v, tag, err)
            np.testing.assert_array_equal(got, gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=True))
            os.environ["DEMON_FORCE_PLAN"] = "16,%d,1" % v
            plain = gpu_ctx.conv2d(x, w, b, (sh, sw), lrelu=True)
            if gpu_ctx.last_kernel() == "wino4<t%d,v%d>" % (max(kh, kw), v):
                np.testing.assert_array_equal(got, plain)   # same template instance, same sums per output: bit-identical
        if (H, W) in ((12, 16), (12, 32), (6, 8), (6, 16)):
            assert flat >= 1, layer
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


@pytest.mark.parametrize("shape", [(24, 4, 48, 64), (24, 4, 21, 48), (16, 1, 40, 192), (16, 1, 9, 200), (24, 3, 17, 130), (10, 2, 12, 36), (24, 4, 48, 256)])
def test_small_heads_narrow_and_wide_tiles(gpu_ctx, shape):
    """conv_small.hip (plan kind 3): the Cout <= 4 heads.  Round 6 gave it 64-wide tiles for maps whose last 128-wide tile would be at most
    half full (the 48 x 64 heads of every block); both tile widths, widths that are not multiples of 4 (scalar staging path), every
    Cout instantiation -- against PyTorch, deterministic."""
    cin, cout, H, W = shape
    rng = np.random.default_rng(61)
    x = rng.standard_normal((3, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("conv", x, w, b, (1, 1))
    try:
        os.environ["DEMON_FORCE_PLAN"] = "3,0,0"
        got = gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True)
        assert gpu_ctx.last_kernel().startswith("conv_small"), gpu_ctx.last_kernel()
        assert rel_l1(got, want) < 1e-5
        np.testing.assert_array_equal(got, gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True))
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)


@pytest.mark.parametrize("layer", [(64, 64, 3, 1, 48, 64), (64, 64, 1, 3, 48, 64), (128, 128, 3, 1, 24, 32), (128, 128, 1, 3, 24, 32), (256, 256, 3, 1, 12, 16)])
def test_tile_walking_workgroups_equal_the_plain_launch(gpu_ctx, layer):
    """conv_wino4.hip, plan field ksplit = 2 (round 6): fewer workgroups than tiles, each walking a whole number of tiles with the next
    tile's first loads issued under the current tile's epilogue.  At a batch where the tiles exceed one round of the chip the walking
    form must really run (kernel tag ",walk"): against PyTorch <= 1e-5, within 1e-6 of the plain launch, deterministic."""
    cin, cout, kh, kw, H, W = layer
    rng = np.random.default_rng(62)
    n = 26
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    b = rng.standard_normal((cout,)).astype(np.float32)
    want = _ref("conv", x, w, b, (1, 1))
    walked = 0
    try:
        for v in range(14):
            os.environ["DEMON_FORCE_PLAN"] = "16,%d,1" % v
            plain = gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True)
            if not gpu_ctx.last_kernel().startswith("wino4<"):
                continue
            assert rel_l1(plain, want) < 1e-5, v
            os.environ["DEMON_FORCE_PLAN"] = "16,%d,2" % v
            got = gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True)
            tag = gpu_ctx.last_kernel()
            assert rel_l1(got, want) < 1e-5 and rel_l1(got, plain) < 1e-6, tag     # (last-bit differences: another template instance, other multiply-add contractions)
            np.testing.assert_array_equal(got, gpu_ctx.conv2d(x, w, b, (1, 1), lrelu=True), err_msg=tag)   # deterministic
            walked += ",walk" in tag
        if H * W >= 24 * 32:   # (the 12 x 16 layer has fewer tiles than the chip has workgroup slots at this batch: nothing to walk, the plain launch runs)
            assert walked >= 2, "no shape ran its tile-walking form at batch %d" % n
    finally:
        os.environ.pop("DEMON_FORCE_PLAN", None)
