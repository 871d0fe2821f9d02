"""`lmbspecialops` for the TensorFlow-1.4 graph emulator next door (oracle/tf1/tensorflow) -- test infrastructure, never imported
by the product path.

The reference loads lmbspecialops (an empty submodule in /root/reference, `.gitmodules:1-3`) with tf.load_op_library and calls
`sops.depth_to_flow / flow_to_depth / flow_to_depth2 / warp2d / leaky_relu / median3x3_downsample / ...` on NCHW tensors
(python/depthmotionnet/blocks_original.py:155-176, :336-351, helpers.py:60-63; v2/blocks.py:154-161, :362-379).  Here every op is a
graph node whose value comes from oracle/ops_ref.py, i.e. from this repo's restatement of the op: the emulator pins how the
reference WIRES these ops (argument order, keyword names, which tensor feeds which), not what the ops compute ("parity
unpinned" for their semantics).  Keyword names are lmbspecialops' own (`input`, `displacements`,
`normalized`, `border_mode`, `normalize_flow`, `normalized_flow`, `inverse_depth`, `rotation_format`, `leak`); unknown keywords raise.
"""
import numpy as np
import tensorflow as tf

if not getattr(tf, "EMULATED", False):
    raise ImportError("oracle/tf1/lmbspecialops is the companion of the TensorFlow emulator oracle/tf1/tensorflow only")

__version__ = "emulated-by-oracle/ops_ref.py"


def _ops():
    from oracle import ops_ref
    return ops_ref


def _node(fn, inputs, shape):
    return tf.Tensor(fn, [tf._as_tensor(t) for t in inputs], shape, tf.float32)


def _rotation_format(fmt):
    if fmt != "angleaxis3":
        raise NotImplementedError("emulator: rotation_format %r (the reference's inference graphs use the default)" % (fmt,))


def leaky_relu(input, leak=0.1, name=None):   # noqa: A002
    input = tf._as_tensor(input)   # noqa: A001
    return _node(lambda x: _ops().leaky_relu(x, leak), [input], input.get_shape().as_list())


def replace_nonfinite(input, value=0.0, name=None):   # noqa: A002
    input = tf._as_tensor(input)   # noqa: A001
    return _node(lambda x: _ops().replace_nonfinite(x, value), [input], input.get_shape().as_list())


def depth_to_flow(depth, intrinsics, rotation, translation, rotation_format="angleaxis3", inverse_depth=False, normalize_flow=False, name=None):
    _rotation_format(rotation_format)
    depth = tf._as_tensor(depth)
    n, c, h, w = depth.get_shape().as_list()
    if c != 1:
        raise ValueError("depth_to_flow: depth must be [N,1,H,W] (NCHW), got %r" % ([n, c, h, w],))
    return _node(lambda d, k, r, t: _ops().depth_to_flow(d, k, r, t, inverse_depth, normalize_flow), [depth, intrinsics, rotation, translation], (n, 2, h, w))


def _f2d(method):
    def op(flow, intrinsics, rotation, translation, rotation_format="angleaxis3", inverse_depth=False, normalized_flow=False, name=None):
        _rotation_format(rotation_format)
        flow = tf._as_tensor(flow)
        n, c, h, w = flow.get_shape().as_list()
        if c != 2:
            raise ValueError("flow_to_depth: flow must be [N,2,H,W] (NCHW), got %r" % ([n, c, h, w],))
        return _node(lambda f, k, r, t: _ops().flow_to_depth(f, k, r, t, inverse_depth, normalized_flow, method), [flow, intrinsics, rotation, translation], (n, 1, h, w))
    return op


flow_to_depth = _f2d(0)
flow_to_depth2 = _f2d(1)


def warp2d(input, displacements, normalized=False, border_mode="clamp", border_value=0.0, name=None):   # noqa: A002
    input, displacements = tf._as_tensor(input), tf._as_tensor(displacements)   # noqa: A001
    shp, dshp = input.get_shape().as_list(), displacements.get_shape().as_list()
    if dshp != [shp[0], 2, shp[2], shp[3]]:
        raise ValueError("warp2d: displacements %r do not fit input %r (both NCHW)" % (dshp, shp))
    if border_mode not in ("clamp", "value"):
        raise ValueError("warp2d: border_mode %r" % (border_mode,))
    return _node(lambda x, d: _ops().warp2d(x, d, normalized, border_mode, border_value), [input, displacements], shp)


def median3x3_downsample(input, name=None):   # noqa: A002
    input = tf._as_tensor(input)   # noqa: A001
    n, c, h, w = input.get_shape().as_list()
    return _node(lambda x: _ops().median3x3_downsample(x), [input], (n, c, (h + 1) // 2, (w + 1) // 2))


def scale_invariant_gradient(input, deltas=(1,), weights=(1.0,), epsilon=0.001, name=None):   # noqa: A002
    input = tf._as_tensor(input)   # noqa: A001
    shp = input.get_shape().as_list()
    out = np.asarray(_ops().scale_invariant_gradient(np.zeros(shp, np.float32), deltas, weights, epsilon)).shape
    return _node(lambda x: _ops().scale_invariant_gradient(x, deltas, weights, epsilon), [input], out)


def depth_to_normals(depth, intrinsics, inverse_depth=False, name=None):
    depth = tf._as_tensor(depth)
    n, c, h, w = depth.get_shape().as_list()
    return _node(lambda d, k: _ops().depth_to_normals(d, k, inverse_depth), [depth, intrinsics], (n, 3, h, w))
