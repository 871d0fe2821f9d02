"""Variable table of the five DeMoN sub-nets, synthetic weights, and weight file IO.

Names and layouts are the TensorFlow ones a `tf.train.Saver` checkpoint of the reference holds
(examples/example.py:82-83; scopes python/depthmotionnet/networks_original.py:44,50,125,142,227; layer
names python/depthmotionnet/blocks_original.py): `<scope>/<layer>/kernel` + `/bias`,
conv HWIO [kh,kw,Cin,Cout], transposed conv [4,4,Cout,Cin], dense [in,out].

This table is written independently of the C++ topology in csrc/demon_api.hip; a GPU test checks
that the two agree name by name and shape by shape.
"""
import numpy as np


def _sep(scope, name, cin, cout, k):
    # helpers.py:105-153: '<name>y' [k,1,cin,cout] then '<name>x' [1,k,cout,cout]
    return [("%s/%sy" % (scope, name), (k, 1, cin, cout)), ("%s/%sx" % (scope, name), (1, k, cout, cout))]


def _encoder(scope, conv2_out, extra_in, conv5_k):
    v = _sep(scope, "conv1", 6, 32, 9) + _sep(scope, "conv2", 32, conv2_out, 7)
    if extra_in:
        v += _sep(scope, "conv2_extra_inputs", extra_in, 32, 3)
    v += _sep(scope, "conv2_1", 64, 64, 3)
    v += _sep(scope, "conv3", 64, 128, 5) + _sep(scope, "conv3_1", 128, 128, 3)
    v += _sep(scope, "conv4", 128, 256, 5) + _sep(scope, "conv4_1", 256, 256, 3)
    v += _sep(scope, "conv5", 256, 512, conv5_k) + _sep(scope, "conv5_1", 512, 512, 3)
    return v


def _flow_net(scope, iterative):
    # blocks_original.py:121-235
    v = _encoder(scope, 32 if iterative else 64, 9 if iterative else 0, 5)
    v += [(scope + "/predict_flow5/conv1", (3, 3, 512, 24)), (scope + "/predict_flow5/conv2", (3, 3, 24, 4)),
          (scope + "/upsample_flow5to4/upconv", (4, 4, 2, 4)),
          (scope + "/refine4/upconv", (4, 4, 256, 512)), (scope + "/refine3/upconv", (4, 4, 128, 514)),
          (scope + "/refine2/upconv", (4, 4, 64, 256)),
          (scope + "/predict_flow2/conv1", (3, 3, 128, 24)), (scope + "/predict_flow2/conv2", (3, 3, 24, 4))]
    return v


def _dm_net(scope, iterative, fc_in):
    # blocks_original.py:299-448
    v = _encoder(scope, 32, 8 if iterative else 7, 3)
    v += [(scope + "/motion_conv1", (3, 3, 512, 128)), (scope + "/motion_fc1", (fc_in, 1024)),
          (scope + "/motion_fc2", (1024, 128)), (scope + "/motion_fc3", (128, 7)),
          (scope + "/refine4/upconv", (4, 4, 256, 512)), (scope + "/refine3/upconv", (4, 4, 128, 512)),
          (scope + "/refine2/upconv", (4, 4, 64, 256)),
          (scope + "/predict_depthnormal2/conv1", (3, 3, 128, 24)),
          (scope + "/predict_depthnormal2/conv2", (3, 3, 24, 4))]
    return v


def _refine_net(scope="netRefine"):
    # blocks_original.py:452-513
    return [(scope + "/conv0", (3, 3, 4, 32)), (scope + "/conv1", (3, 3, 32, 64)), (scope + "/conv1_1", (3, 3, 64, 64)),
            (scope + "/conv2", (3, 3, 64, 128)), (scope + "/conv2_1", (3, 3, 128, 128)),
            (scope + "/refine1/upconv", (4, 4, 64, 128)), (scope + "/refine0/upconv", (4, 4, 32, 128)),
            (scope + "/predict_depth0/conv1", (3, 3, 64, 16)), (scope + "/predict_depth0/conv2", (3, 3, 16, 1))]


# ---- v2 (python/depthmotionnet/v2/blocks.py) ------------------------------------------------------------
def _sep2(scope, name, cin, cout, k):
    # v2/helpers.py:44-91: num_outputs may be (outputs of the k x 1 filter, outputs of the 1 x k filter)
    cy, cx = cout if isinstance(cout, tuple) else (cout, cout)
    return [("%s/%sy" % (scope, name), (k, 1, cin, cy)), ("%s/%sx" % (scope, name), (1, k, cy, cx))]


def _encoder_v2(scope, conv2_out, extra_in, conv5_k, hw5):
    # v2/blocks.py:141-213 (flow), :344-411 (depth+motion)
    v = _sep2(scope, "conv1", 6, (24, 32), 9) + _sep2(scope, "conv2", 32, conv2_out, 7)
    if extra_in:
        v += _sep2(scope, "conv2_extra_inputs", extra_in, 32, 3)
    v += _sep2(scope, "conv2_1", 64, 64, 3)
    v += _sep2(scope, "conv3", 64, (96, 128), 5) + _sep2(scope, "conv3_1", 128, 128, 3)
    v += _sep2(scope, "conv4", 128, (192, 256), 5) + _sep2(scope, "conv4_1", 256, 256, 3)
    v += _sep2(scope, "conv5", 256, 384, conv5_k) + _sep2(scope, "conv5_1", 384, 384, 3)
    v += [(scope + "/dense5", (96 * hw5, 96 * hw5))]
    return v


def _flow_net_v2(scope, iterative, hw5):
    # v2/blocks.py:120-253
    v = _encoder_v2(scope, 32 if iterative else (48, 64), 9 if iterative else 0, 5, hw5)
    v += [(scope + "/predict_flow5/conv1", (3, 3, 480, 24)), (scope + "/predict_flow5/conv2", (3, 3, 24, 4)),
          (scope + "/upsample_flow5to4/upconv", (4, 4, 2, 4)),
          (scope + "/refine4/upconv", (4, 4, 256, 480)), (scope + "/refine3/upconv", (4, 4, 128, 514)),
          (scope + "/refine2/upconv", (4, 4, 64, 256)),
          (scope + "/predict_flow2/conv1", (3, 3, 128, 24)), (scope + "/predict_flow2/conv2", (3, 3, 24, 4))]
    return v


def _dm_net_v2(scope, iterative, hw5):
    # v2/blocks.py:317-494
    v = _encoder_v2(scope, 32, 8 if iterative else 7, 3, hw5)
    v += _sep2(scope, "motion_conv3", 64, 64, 5) + _sep2(scope, "motion_conv4", 64, 64, 5)
    v += _sep2(scope, "motion_conv5a", 64, 64, 3)
    v += [(scope + "/motion_conv5b", (3, 3, 480, 64)), (scope + "/motion_fc1", (128 * hw5, 1024)),
          (scope + "/motion_fc2", (1024, 128)), (scope + "/motion_fc3", (128, 7)),
          (scope + "/refine4/upconv", (4, 4, 256, 384)), (scope + "/refine3/upconv", (4, 4, 128, 512)),
          (scope + "/refine2/upconv", (4, 4, 64, 256)),
          (scope + "/predict_depthnormal2/conv1", (3, 3, 128, 24)),
          (scope + "/predict_depthnormal2/conv2", (3, 3, 24, 4))]
    return v


def _refine_net_v2(scope="netRefine"):
    # v2/blocks.py:499-562: as the original but the head predicts depth + normal (4 channels)
    return _refine_net(scope)[:-1] + [(scope + "/predict_depth0/conv2", (3, 3, 16, 4))]


def layer_table(height=192, width=256, version=1):
    """[(layer_name, kernel_shape)]: 121 layers for the original model, 137 for v2."""
    hw5 = (height // 32) * (width // 32)
    if version == 2:
        return (_flow_net_v2("netFlow1", False, hw5) + _dm_net_v2("netDM1", False, hw5) +
                _flow_net_v2("netFlow2", True, hw5) + _dm_net_v2("netDM2", True, hw5) + _refine_net_v2())
    fc_in = 128 * hw5  # 6144 at 192x256 (blocks_original.py:380-396)
    return (_flow_net("netFlow1", False) + _dm_net("netDM1", False, fc_in) + _flow_net("netFlow2", True) +
            _dm_net("netDM2", True, fc_in) + _refine_net())


def weights_version(weights):
    """2 if the dict holds the v2 model's variables (it has a dense5 layer), else 1."""
    return 2 if "netFlow1/dense5/kernel" in weights else 1


def _cout(name, shape):
    return shape[2] if name.endswith("upconv") else shape[-1]


def variable_shapes(height=192, width=256, version=1):
    """dict tf variable name -> shape (original: 242 tensors, 45 753 883 floats at 192x256)."""
    out = {}
    for name, shape in layer_table(height, width, version):
        out[name + "/kernel"] = tuple(shape)
        out[name + "/bias"] = (_cout(name, shape),)
    return out


def blob_order(height=192, width=256, version=1):
    """[(variable name, shape)] in the order of the flat weight blob = the order libdemon_hip.so creates its variables
    (demon_variable_info / DemonContext.variables(); a GPU test holds the two lists equal, element by element)."""
    return list(variable_shapes(height, width, version).items())


def blob_to_weights(blob, order):
    """inverse of weights_to_blob: views into `blob`"""
    w, off = {}, 0
    for name, shape in order:
        cnt = int(np.prod(shape))
        w[name] = blob[off:off + cnt].reshape(shape)
        off += cnt
    return w


_LINEAR = ("predict_flow5/conv2", "predict_flow2/conv2", "predict_depthnormal2/conv2", "predict_depth0/conv2",
           "upsample_flow5to4/upconv", "motion_fc3")


def synthetic_weights(seed=1, height=192, width=256, head_scale=0.1, version=1, consistent_flow=None):
    """Deterministic random weights that keep the nets in their working regime.

    kernels ~ N(0, 2/fan_in) (He, the reference's variance_scaling_initializer, helpers.py:66-67), with the
    effective fan-in of the 2x2 sub-pixel taps for the transposed convs; biases ~ N(0, 0.01^2); the linear
    heads are scaled by `head_scale` so that |flow| < 1 for most pixels; the motion head is biased towards
    t = (0.8, 0.2, -0.1), scale = 1 and the depth head towards inverse depth 0.5 so that depth_to_flow /
    flow_to_depth see valid geometry (otherwise the NaN gate of blocks_original.py:163-168 would be all
    that is tested).  head_scale=1.0 gives the "gate stress" variant.

    consistent_flow (default: on for version 2): the level-2 flow head is biased towards the flow that the biased motion and
    depth imply (u ~ fx*tx*d = 0.356, v ~ fy*ty*d = 0.119, normalized).  The v2 depth+motion block feeds
    clip(1 / flow_to_depth2(flow), 0, 50) to a conv (v2/blocks.py:362-381): where the triangulated depth passes through zero
    the input jumps between 0 and 50, so with random flows the MODEL is chaotic -- a 1e-7 perturbation of the images moves the
    oracle's own outputs by 1e-2 after three iterations -- and no implementation can be compared at 1e-3.  With flows that
    agree with the motion the triangulated depth stays positive and the same perturbation moves the outputs by 1e-6.
    consistent_flow=False gives the "clip stress" variant.
    """
    if consistent_flow is None:
        consistent_flow = version == 2
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in layer_table(height, width, version):
        if name.endswith("upconv"):
            fan_in = 4 * shape[3]
        elif len(shape) == 2:
            fan_in = shape[0]
        else:
            fan_in = shape[0] * shape[1] * shape[2]
        k = rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
        b = rng.standard_normal((_cout(name, shape),), dtype=np.float32) * np.float32(0.01)
        if any(name.endswith(s) for s in _LINEAR):
            k *= np.float32(head_scale)
        if name.endswith("motion_fc3"):
            b += np.array([0.0, 0.0, 0.0, 0.8, 0.2, -0.1, 1.0], np.float32)
        if consistent_flow and name.endswith("predict_flow2/conv2"):
            b[0:2] += np.array([0.356, 0.119], np.float32)
        if name.endswith("predict_depthnormal2/conv2"):
            b[0] += np.float32(0.5)
        if name.endswith("predict_depth0/conv2"):
            b[0] += np.float32(0.5)
        w[name + "/kernel"] = k
        w[name + "/bias"] = b
    return w


def weights_to_blob(weights, order):
    """Flattens a weights dict into one float32 vector following `order` = [(name, shape)]."""
    return np.concatenate([np.ascontiguousarray(weights[n], np.float32).reshape(-1) for n, _ in order])


def save_npz(path, weights):
    np.savez(path, **{k.replace("/", "."): v for k, v in weights.items()})


def load_npz(path):
    with np.load(path) as f:
        return {k.replace(".", "/"): f[k] for k in f.files}
