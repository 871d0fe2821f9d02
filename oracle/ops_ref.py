"""ctypes front end of oracle/demon_oracle.c (numpy in, numpy out).  TEST INFRASTRUCTURE ONLY.

Each function mirrors one lmbspecialops / TensorFlow call made by the reference; the C file cites the
reference file:line it follows.  All arrays are NCHW float32.
"""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def lib():
    global _lib
    if _lib is None:
        path = _build.OUT
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(_build.SRC):
            _build.build()
        _lib = ctypes.CDLL(path)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def angleaxis_to_rotation(aa):
    aa, pa = _f(aa)
    R = np.empty((3, 3), np.float32)
    lib().ref_angleaxis_to_rotation(pa, R.ctypes.data_as(_f32p))
    return R


def depth_to_flow(depth, intrinsics, rotation, translation, inverse_depth=False, normalize_flow=False, gate=False):
    depth, pd = _f(depth)
    N, C, H, W = depth.shape
    assert C == 1
    intrinsics, pi = _f(np.broadcast_to(intrinsics, (N, 4)))
    rotation, pr = _f(rotation)
    translation, pt = _f(translation)
    out = np.empty((N, 2, H, W), np.float32)
    lib().ref_depth_to_flow(out.ctypes.data_as(_f32p), pd, pi, pr, pt, N, H, W, int(inverse_depth),
                            int(normalize_flow), int(gate))
    return out


def flow_to_depth(flow, intrinsics, rotation, translation, inverse_depth=False, normalized_flow=False, method=0):
    flow, pf = _f(flow)
    N, C, H, W = flow.shape
    assert C == 2
    intrinsics, pi = _f(np.broadcast_to(intrinsics, (N, 4)))
    rotation, pr = _f(rotation)
    translation, pt = _f(translation)
    out = np.empty((N, 1, H, W), np.float32)
    lib().ref_flow_to_depth(out.ctypes.data_as(_f32p), pf, pi, pr, pt, N, H, W, int(inverse_depth),
                            int(normalized_flow), int(method))
    return out


def flow_to_depth2(flow, intrinsics, rotation, translation, inverse_depth=False, normalized_flow=False):
    return flow_to_depth(flow, intrinsics, rotation, translation, inverse_depth, normalized_flow, method=1)


def warp2d(inp, displacements, normalized=False, border_mode="clamp", border_value=0.0):
    inp, pi = _f(inp)
    displacements, pd = _f(displacements)
    N, C, H, W = inp.shape
    assert displacements.shape == (N, 2, H, W)
    out = np.empty_like(inp)
    lib().ref_warp2d(out.ctypes.data_as(_f32p), pi, pd, N, C, H, W, int(normalized),
                     1 if border_mode == "value" else 0, ctypes.c_float(border_value))
    return out


def leaky_relu(x, leak=0.1):
    x, px = _f(x)
    out = np.empty_like(x)
    lib().ref_leaky_relu(out.ctypes.data_as(_f32p), px, ctypes.c_size_t(x.size), ctypes.c_float(leak))
    return out


def replace_nonfinite(x, value=0.0):
    x, px = _f(x)
    out = np.empty_like(x)
    lib().ref_replace_nonfinite(out.ctypes.data_as(_f32p), px, ctypes.c_size_t(x.size), ctypes.c_float(value))
    return out


def scale_invariant_gradient(x, deltas=(1,), weights=(1.0,), epsilon=0.001):
    x, px = _f(x)
    N, C, H, W = x.shape
    d = np.ascontiguousarray(deltas, np.int32)
    w, pw = _f(weights)
    out = np.empty((N * C, 2, H, W), np.float32)   # channels fold into the batch, deltas are summed (lmbspecialops contract)
    lib().ref_scale_invariant_gradient(out.ctypes.data_as(_f32p), px, N * C, H, W, d.ctypes.data_as(_i32p), pw,
                                       len(d), ctypes.c_float(epsilon))
    return out


def depth_to_normals(depth, intrinsics, inverse_depth=False):
    depth, pd = _f(depth)
    N, C, H, W = depth.shape
    assert C == 1
    intrinsics, pi = _f(np.broadcast_to(intrinsics, (N, 4)))
    out = np.empty((N, 3, H, W), np.float32)
    lib().ref_depth_to_normals(out.ctypes.data_as(_f32p), pd, pi, N, H, W, int(inverse_depth))
    return out


def median3x3_downsample(x):
    x, px = _f(x)
    N, C, H, W = x.shape
    out = np.empty((N, C, (H + 1) // 2, (W + 1) // 2), np.float32)
    lib().ref_median3x3_downsample(out.ctypes.data_as(_f32p), px, N * C, H, W)
    return out


def conv2d_hwio(x, w, b, stride, pad, lrelu):
    x, px = _f(x)
    w, pw = _f(w)
    b, pb = _f(b)
    N, Cin, H, W = x.shape
    kh, kw, ci, Cout = w.shape
    assert ci == Cin
    sh, sw = stride
    ph, pw_ = pad
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw_ - kw) // sw + 1
    out = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().ref_conv2d_hwio(out.ctypes.data_as(_f32p), px, pw, pb, N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw_,
                          int(lrelu))
    return out


def conv2d_hwio_same(x, w, b, stride, lrelu):
    """tf.layers.conv2d(padding='same') (v2/helpers.py:24-35) by the naive loops: the input is zero-padded explicitly with
    TensorFlow's rule (out = ceil(n/s); pad_total = max((out-1)*s + k - n, 0); pad_total // 2 in front) and convolved VALID."""
    x = np.asarray(x, np.float32)
    kh, kw = w.shape[0], w.shape[1]
    pads = []
    for n, k, s in ((x.shape[2], kh, stride[0]), (x.shape[3], kw, stride[1])):
        out = -(-n // s)
        total = max((out - 1) * s + k - n, 0)
        pads.append((total // 2, total - total // 2))
    xp = np.pad(x, ((0, 0), (0, 0), pads[0], pads[1]))
    return conv2d_hwio(xp, w, b, stride, (0, 0), lrelu)


def deconv4x4s2_crop(x, w, b, lrelu):
    x, px = _f(x)
    w, pw = _f(w)
    b, pb = _f(b)
    N, Cin, H, W = x.shape
    assert w.shape[0] == 4 and w.shape[1] == 4 and w.shape[3] == Cin
    Cout = w.shape[2]
    out = np.empty((N, Cout, 2 * H, 2 * W), np.float32)
    lib().ref_deconv4x4s2_crop(out.ctypes.data_as(_f32p), px, pw, pb, N, Cin, H, W, Cout, int(lrelu))
    return out


def dense(x, w, b, lrelu):
    x, px = _f(x)
    w, pw = _f(w)
    b, pb = _f(b)
    N, Cin = x.shape
    Cout = w.shape[1]
    out = np.empty((N, Cout), np.float32)
    lib().ref_dense(out.ctypes.data_as(_f32p), px, pw, pb, N, Cin, Cout, int(lrelu))
    return out


def resize_nearest(x, Ho, Wo):
    x, px = _f(x)
    N, C, H, W = x.shape
    out = np.empty((N, C, Ho, Wo), np.float32)
    lib().ref_resize_nearest(out.ctypes.data_as(_f32p), px, N * C, H, W, Ho, Wo)
    return out


def pointwise_l2_loss(inp, gt, epsilon):
    """python/depthmotionnet/v2/losses.py:33-54 (NCHW): diff = replace_nonfinite(inp - gt);
    reduce_mean(sqrt(reduce_sum(diff**2, axis=1) + epsilon)).  numpy, float64 accumulation."""
    with np.errstate(invalid="ignore"):
        d = np.asarray(inp, np.float32) - np.asarray(gt, np.float32)
    d = np.where(np.isfinite(d), d, np.float32(0)).astype(np.float64)
    return float(np.sqrt((d * d).sum(axis=1) + epsilon).mean())
