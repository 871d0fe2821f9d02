"""Shared between tools/dump_reference_goldens.py (runs in the REFERENCE's environment: TensorFlow 1.4 + lmbspecialops, Python 3.5,
numpy 1.13 -- Dockerfile:14-27 of lmb-freiburg/demon) and tests/test_tf_goldens.py (runs here).  numpy only, no f-strings, no
numpy.random.Generator: everything random comes from numpy.random.RandomState, whose streams are frozen across numpy versions,
so both sides can regenerate the same weights and inputs from a seed and the golden files only have to carry the outputs.

Nothing in here is product code; it is test infrastructure like oracle/.
"""
import json
import zlib

import numpy as np

FORMAT_VERSION = 1
INTRINSICS = (0.89115971, 1.18821287, 0.5, 0.5)   # networks_original.py:108-109


def _rs(seed, name):
    """one independent stream per (seed, name): the order and the set of variables do not matter"""
    return np.random.RandomState((zlib.crc32(name.encode("utf-8")) ^ (seed * 2654435761)) & 0xFFFFFFFF)


_LINEAR = ("predict_flow5/conv2", "predict_flow2/conv2", "predict_depthnormal2/conv2", "predict_depth0/conv2",
           "upsample_flow5to4/upconv", "motion_fc3")


def seeded_weights(var_shapes, seed=1, head_scale=0.1, consistent_flow=False):
    """var_shapes: iterable of (tf variable name "scope/layer/kernel" | ".../bias", shape).  The recipe of
    demon_amd.weights.synthetic_weights (He-normal kernels, N(0, 0.01^2) biases, linear heads scaled by head_scale, motion / depth
    heads biased into the working regime), drawn from RandomState so that the reference's environment reproduces it bit for bit."""
    out = {}
    for name, shape in var_shapes:
        shape = tuple(int(s) for s in shape)
        layer = name.rsplit("/", 1)[0]
        rs = _rs(seed, name)
        if name.endswith("/kernel"):
            if layer.endswith("upconv"):
                fan_in = 4 * shape[3]                       # [4,4,Cout,Cin]: 2 x 2 taps reach an output pixel
            elif len(shape) == 2:
                fan_in = shape[0]
            else:
                fan_in = shape[0] * shape[1] * shape[2]
            k = (rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
            if any(layer.endswith(s) for s in _LINEAR):
                k = (k * np.float32(head_scale)).astype(np.float32)
            out[name] = k
        else:
            b = (rs.standard_normal(shape) * 0.01).astype(np.float32)
            if layer.endswith("motion_fc3"):
                b = b + np.array([0.0, 0.0, 0.0, 0.8, 0.2, -0.1, 1.0], np.float32)
            if consistent_flow and layer.endswith("predict_flow2/conv2"):
                b[0:2] += np.array([0.356, 0.119], np.float32)
            if layer.endswith("predict_depthnormal2/conv2") or layer.endswith("predict_depth0/conv2"):
                b[0] += np.float32(0.5)
            out[name] = b.astype(np.float32)
    return out


def synthetic_pair(n, seed, height=192, width=256):
    rs = np.random.RandomState(seed)
    pair = (rs.random_sample((n, 6, height, width)) - 0.5).astype(np.float32)
    img2_2 = pair[:, 3:6].reshape(n, 3, height // 4, 4, width // 4, 4).mean(axis=(3, 5)).astype(np.float32)
    return pair, img2_2


# ---- geometry helper used only to BUILD inputs (a flow field that is consistent with a depth map and a motion) ----------------
def _rodrigues(aa):
    aa = np.asarray(aa, np.float64)
    angle = np.linalg.norm(aa)
    if angle <= 1e-6:
        return np.eye(3)
    k = aa / angle
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K.dot(K)


def np_depth_to_flow(depth, intrinsics, rotation, translation, normalize=True):
    """depth [N,1,H,W] (depth, not inverse) -> flow [N,2,H,W]: pixel centres at +0.5, X2 = R X1 + t, flow = p2 - p1
    (dataset_tools/view_tools_cython.pyx:196-240).  float64, for input construction only."""
    n, _, h, w = depth.shape
    out = np.zeros((n, 2, h, w), np.float64)
    ys, xs = np.mgrid[0:h, 0:w]
    for i in range(n):
        fx, fy, cx, cy = [float(v) for v in intrinsics[i]]
        fx, fy, cx, cy = fx * w, fy * h, cx * w, cy * h
        R, t = _rodrigues(rotation[i]), np.asarray(translation[i], np.float64)
        d = depth[i, 0].astype(np.float64)
        X = np.stack([(xs + 0.5 - cx) / fx * d, (ys + 0.5 - cy) / fy * d, d], 0).reshape(3, -1)
        X2 = R.dot(X) + t[:, None]
        px = fx * X2[0] / X2[2] + cx
        py = fy * X2[1] / X2[2] + cy
        out[i, 0] = px.reshape(h, w) - (xs + 0.5)
        out[i, 1] = py.reshape(h, w) - (ys + 0.5)
        if normalize:
            out[i, 0] /= w
            out[i, 1] /= h
    return out.astype(np.float32)


# ---- op cases: {"op", "kwargs", "inputs" (ordered list of (argument name, array))} -------------------------------------------
def op_cases(seed=7):
    rs = np.random.RandomState(seed)
    cases = []
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    intr = np.tile(np.array(INTRINSICS, np.float32), (2, 1))
    rot = np.array([[0.02, -0.05, 0.01], [0.0, 0.0, 0.0]], np.float32)              # second sample: angle below 1e-6 -> identity
    rot_big = np.array([[0.4, -0.9, 0.3], [1e-7, 0.0, 0.0]], np.float32)
    trans = np.array([[0.8, 0.2, -0.1], [-0.3, 0.1, 0.9]], np.float32)
    h, w = 12, 16

    # -- warp2d (blocks_original.py:171-176, :336): taps outside the image, on the border, non-finite displacements
    img = rs.standard_normal((2, 3, h, w)).astype(np.float32)
    small = (rs.standard_normal((2, 2, h, w)) * 0.7).astype(np.float32)
    big = (rs.standard_normal((2, 2, h, w)) * 9.0).astype(np.float32)
    edge = np.zeros((2, 2, h, w), np.float32)
    edge[0, 0, :, :4] = -1.0; edge[0, 0, :, 4:8] = -0.5; edge[0, 0, :, 8:12] = -1.5; edge[0, 0, :, 12:] = 0.25      # x: left of / on the first column
    edge[0, 1, :3, :] = -1.0; edge[0, 1, 3:6, :] = -3.5; edge[0, 1, 9:, :] = 2.0                                    # y: above / below
    edge[1, 0] = np.float32(w) - 1.0 - np.arange(w, dtype=np.float32)[None, :]                                        # every pixel samples the last column ...
    edge[1, 0, 6:, :] += 0.5                                                                                           # ... or half a pixel beyond it
    edge[1, 1] = (np.float32(h) - 1.0 - np.arange(h, dtype=np.float32))[:, None]
    bad = small.copy()
    bad[0, 0, 2, 3] = nan; bad[0, 1, 5, 7] = nan; bad[1, 0, 0, 0] = inf; bad[1, 1, 11, 15] = -inf; bad[1, :, 6, 6] = nan
    for name, disp in (("small", small), ("big", big), ("edge", edge), ("nonfinite", bad)):
        for border_mode in ("value", "clamp"):
            cases.append({"op": "warp2d", "tag": "warp2d/%s/%s/pixels" % (name, border_mode),
                          "inputs": [("input", img), ("displacements", disp)],
                          "kwargs": {"normalized": False, "border_mode": border_mode}})
        norm = disp / np.array([w, h], np.float32).reshape(1, 2, 1, 1)
        cases.append({"op": "warp2d", "tag": "warp2d/%s/value/normalized" % name,
                      "inputs": [("input", img), ("displacements", norm.astype(np.float32))],
                      "kwargs": {"normalized": True, "border_mode": "value"}})
    cases.append({"op": "warp2d", "tag": "warp2d/big/value0.5", "inputs": [("input", img), ("displacements", big)],
                  "kwargs": {"normalized": False, "border_mode": "value", "border_value": 0.5}})

    # -- depth_to_flow (blocks_original.py:155-162; v2/losses.py:332; evaluation.py:81)
    depth = (0.5 + rs.random_sample((2, 1, h, w)) * 3.0).astype(np.float32)
    depth_bad = depth.copy()
    depth_bad[0, 0, 0, 0] = 0.0; depth_bad[0, 0, 1, 1] = -1.0; depth_bad[0, 0, 2, 2] = nan; depth_bad[1, 0, 3, 3] = inf; depth_bad[1, 0, 4, 4] = 1e-12
    for dname, d in (("valid", depth), ("invalid", depth_bad)):
        for rname, r in (("small", rot), ("big", rot_big)):
            for inverse in (False, True):
                for normalize in (False, True):
                    cases.append({"op": "depth_to_flow", "tag": "depth_to_flow/%s/%s/inv%d/norm%d" % (dname, rname, inverse, normalize),
                                  "inputs": [("depth", d), ("intrinsics", intr), ("rotation", r), ("translation", trans)],
                                  "kwargs": {"inverse_depth": bool(inverse), "normalize_flow": bool(normalize)}})

    # -- flow_to_depth / flow_to_depth2 (blocks_original.py:344-360, v2/blocks.py:362): consistent, perturbed and arbitrary flow
    consistent = np_depth_to_flow(depth, intr, rot, trans, normalize=True)
    perturbed = (consistent + rs.standard_normal(consistent.shape) * 0.01).astype(np.float32)
    arbitrary = (rs.standard_normal(consistent.shape) * 0.1).astype(np.float32)
    zero = np.zeros_like(consistent)
    nf = perturbed.copy()
    nf[0, 0, 1, 1] = nan; nf[1, 1, 2, 2] = inf
    for op in ("flow_to_depth", "flow_to_depth2"):
        for fname, f in (("consistent", consistent), ("perturbed", perturbed), ("arbitrary", arbitrary), ("zero", zero), ("nonfinite", nf)):
            for inverse in (False, True):
                cases.append({"op": op, "tag": "%s/%s/inv%d/normalized" % (op, fname, inverse),
                              "inputs": [("flow", f), ("intrinsics", intr), ("rotation", rot), ("translation", trans)],
                              "kwargs": {"inverse_depth": bool(inverse), "normalized_flow": True}})
        px = (perturbed * np.array([w, h], np.float32).reshape(1, 2, 1, 1)).astype(np.float32)
        cases.append({"op": op, "tag": "%s/perturbed/inv1/pixels" % op,
                      "inputs": [("flow", px), ("intrinsics", intr), ("rotation", rot), ("translation", trans)],
                      "kwargs": {"inverse_depth": True, "normalized_flow": False}})

    # -- scale_invariant_gradient (v2/losses.py:58-79): borders (delta up to 16 on a 20 x 24 map), zeros, non-finite values
    x = rs.standard_normal((2, 2, 20, 24)).astype(np.float32)
    x[0, 0, 3:6, 3:6] = 0.0
    xb = x.copy()
    xb[0, 1, 10, 10] = nan; xb[1, 0, 0, 0] = inf
    for xname, xx in (("finite", x), ("nonfinite", xb)):
        for deltas, weights in (([1], [1.0]), ([2], [1.0]), ([4], [0.5]), ([16], [1.0]), ([1, 2], [1.0, 2.0])):
            cases.append({"op": "scale_invariant_gradient", "tag": "sig/%s/d%s" % (xname, "_".join(str(d) for d in deltas)),
                          "inputs": [("input", xx)], "kwargs": {"deltas": deltas, "weights": weights, "epsilon": 0.001}})

    # -- depth_to_normals (v2/losses.py:336-337)
    ys, xs = np.mgrid[0:h, 0:w]
    plane = (2.0 + 0.05 * xs - 0.03 * ys).astype(np.float32)[None, None].repeat(2, 0)
    rough = (1.0 + rs.random_sample((2, 1, h, w))).astype(np.float32)
    rough_bad = rough.copy()
    rough_bad[0, 0, 5, 5] = nan; rough_bad[1, 0, 6, 6] = 0.0
    for dname, d in (("plane", plane), ("rough", rough), ("invalid", rough_bad)):
        for inverse in (False, True):
            cases.append({"op": "depth_to_normals", "tag": "depth_to_normals/%s/inv%d" % (dname, inverse),
                          "inputs": [("depth", d), ("intrinsics", intr)], "kwargs": {"inverse_depth": bool(inverse)}})

    # -- elementwise
    e = rs.standard_normal((2, 3, 5, 7)).astype(np.float32)
    e[0, 0, 0, :3] = [nan, inf, -inf]
    cases.append({"op": "replace_nonfinite", "tag": "replace_nonfinite/default", "inputs": [("input", e)], "kwargs": {}})
    cases.append({"op": "replace_nonfinite", "tag": "replace_nonfinite/value", "inputs": [("input", e)], "kwargs": {"value": -2.5}})
    cases.append({"op": "leaky_relu", "tag": "leaky_relu/0.1", "inputs": [("input", e)], "kwargs": {"leak": 0.1}})

    # -- median3x3_downsample (evaluation.py:173, v2/helpers.py:102): 0 .. 9 NaNs per window, odd sizes
    m = rs.standard_normal((1, 2, 12, 14)).astype(np.float32)
    mn = m.copy()
    k = 0
    for oy in range(0, 12, 2):
        for ox in range(0, 14, 2):
            cnt = k % 10
            k += 1
            cells = [(oy + dy, ox + dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if 0 <= oy + dy < 12 and 0 <= ox + dx < 14]
            for (yy, xx2) in cells[:cnt]:
                mn[0, 0, yy, xx2] = nan
    mo = rs.standard_normal((1, 1, 9, 11)).astype(np.float32)
    for name, arr in (("finite", m), ("nans", mn), ("odd", mo)):
        cases.append({"op": "median3x3_downsample", "tag": "median3x3_downsample/%s" % name, "inputs": [("input", arr)], "kwargs": {}})
    return cases


# ---- layer cases: the TF layer forms of helpers.py / blocks_original.py / v2/helpers.py ----------------------------------------
def layer_cases(seed=11):
    """{"kind", "tag", x [N,C,H,W], params}: kind in convrelu2 (helpers.py:105-153), conv (helpers.py:70-102), conv_same / convrelu2_same
    (v2/helpers.py), upsample_prediction (blocks_original.py:54-75), refine (:79-117), flatten_dense (:380-396)"""
    rs = np.random.RandomState(seed)

    def x(n, c, h, w):
        return rs.standard_normal((n, c, h, w)).astype(np.float32)

    cases = []
    for k, stride, cin, cout, h, w in ((9, 2, 6, 32, 24, 32), (7, 2, 32, 64, 12, 16), (5, 2, 64, 128, 12, 16), (3, 1, 64, 64, 12, 16), (3, 1, 8, 32, 11, 13)):
        cases.append({"kind": "convrelu2", "tag": "convrelu2/k%d_s%d_%dto%d_%dx%d" % (k, stride, cin, cout, h, w), "x": x(2, cin, h, w),
                      "params": {"num_outputs": cout, "kernel_size": k, "stride": stride}})
    for k, stride, cin, cout, h, w, act in ((3, 1, 24, 4, 12, 16, False), (3, 1, 32, 24, 12, 16, True), (3, 2, 32, 64, 24, 32, True), (3, 2, 16, 32, 11, 13, True)):
        cases.append({"kind": "conv", "tag": "conv/k%d_s%d_%dto%d_%dx%d_act%d" % (k, stride, cin, cout, h, w, act), "x": x(2, cin, h, w),
                      "params": {"num_outputs": cout, "kernel_size": k, "strides": stride, "activation": bool(act)}})
    for k, stride, cin, cout, h, w in ((3, 2, 16, 32, 12, 16), (3, 1, 16, 16, 11, 13), (3, 2, 16, 32, 11, 13)):
        cases.append({"kind": "conv_same", "tag": "conv_same/k%d_s%d_%dto%d_%dx%d" % (k, stride, cin, cout, h, w), "x": x(2, cin, h, w),
                      "params": {"num_outputs": cout, "kernel_size": k, "strides": stride}})
    for k, stride, cin, cy, cx, h, w in ((9, 2, 6, 24, 32, 24, 32), (5, 2, 32, 48, 64, 12, 16), (3, 1, 32, 32, 32, 11, 13), (7, 2, 16, 24, 32, 13, 18)):
        cases.append({"kind": "convrelu2_same", "tag": "convrelu2_same/k%d_s%d_%dto%d_%d_%dx%d" % (k, stride, cin, cy, cx, h, w), "x": x(2, cin, h, w),
                      "params": {"num_outputs": [cy, cx], "kernel_size": k, "stride": stride}})
    cases.append({"kind": "upsample_prediction", "tag": "upsample_prediction/4to2_6x8", "x": x(2, 4, 6, 8), "params": {"num_outputs": 2}})
    # features_direct must have num_outputs channels: blocks_original.py:109-110 crops the upconv with tf.slice(tmp, [0,0,1,1],
    # features_direct.get_shape()) -- ALL four dimensions of features_direct, so fewer channels there would also cut the upconv's
    # channels (found by the dry run of the dump tool on oracle/tf1; every refine* of the nets has equal channel counts)
    cases.append({"kind": "refine", "tag": "refine/32to16_6x8", "x": x(2, 32, 6, 8),
                  "params": {"num_outputs": 16}, "features_direct": x(2, 16, 12, 16), "upsampled_prediction": x(2, 2, 12, 16)})
    cases.append({"kind": "refine", "tag": "refine/16to8_5x7_nopred", "x": x(1, 16, 5, 7),
                  "params": {"num_outputs": 8}, "features_direct": x(1, 8, 10, 14), "upsampled_prediction": None})
    cases.append({"kind": "flatten_dense", "tag": "flatten_dense/8x3x4to16", "x": x(3, 8, 3, 4), "params": {"units": 16, "activation": True}})
    return cases


def dumps_kwargs(kwargs):
    return json.dumps(kwargs, sort_keys=True)


def pack_cases(cases, outputs):
    """-> dict of arrays for numpy.savez: case i = keys "<i>/tag", "<i>/op", "<i>/kwargs" (json), "<i>/in/<k>/<argument>", "<i>/out" """
    blob = {"format_version": np.array(FORMAT_VERSION), "count": np.array(len(cases))}
    for i, (c, out) in enumerate(zip(cases, outputs)):
        blob["%d/tag" % i] = np.array(c["tag"])
        blob["%d/op" % i] = np.array(c["op"])
        blob["%d/kwargs" % i] = np.array(dumps_kwargs(c["kwargs"]))
        for k, (name, arr) in enumerate(c["inputs"]):
            blob["%d/in/%d/%s" % (i, k, name)] = np.asarray(arr, np.float32)
        blob["%d/out" % i] = np.asarray(out, np.float32)
    return blob


def unpack_cases(npz):
    """inverse of pack_cases on a loaded npz"""
    out = []
    for i in range(int(npz["count"])):
        ins = sorted((k for k in npz.files if k.startswith("%d/in/" % i)), key=lambda k: int(k.split("/")[2]))
        out.append({"tag": str(npz["%d/tag" % i]), "op": str(npz["%d/op" % i]), "kwargs": json.loads(str(npz["%d/kwargs" % i])),
                    "inputs": [(k.split("/", 3)[3], npz[k]) for k in ins], "out": npz["%d/out" % i]})
    return out
