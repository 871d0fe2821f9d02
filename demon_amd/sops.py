"""Mirror of the `lmbspecialops` Python API (the op set the reference calls) on
numpy arrays, executed by the HIP kernels of libdemon_hip.so.  Argument names / order follow the
reference's call sites (python/depthmotionnet/blocks_original.py:155-176, :336-360; v2/blocks.py:362;
v2/losses.py:49, :78, :336; examples/evaluation.py:173).  All image tensors are NCHW float32.
"""
import numpy as np

from . import runtime


def _ctx():
    return runtime.get_ops_context()


def depth_to_flow(depth, intrinsics, rotation, translation, rotation_format="angleaxis3", inverse_depth=False,
                  normalize_flow=False, name=None):
    """positional order of the reference's positional call sites (examples/evaluation.py:81, v2/losses.py:332-334); the network
    code passes keywords (blocks_original.py:155-162)"""
    if rotation_format != "angleaxis3":
        raise ValueError("only rotation_format='angleaxis3' is supported")
    return _ctx().depth_to_flow(depth, intrinsics, rotation, translation, inverse_depth, normalize_flow)


def flow_to_depth(flow, intrinsics, rotation, translation, rotation_format="angleaxis3", inverse_depth=False,
                  normalized_flow=False):
    if rotation_format != "angleaxis3":
        raise ValueError("only rotation_format='angleaxis3' is supported")
    return _ctx().flow_to_depth(flow, intrinsics, rotation, translation, inverse_depth, normalized_flow, method=0)


def flow_to_depth2(flow, intrinsics, rotation, translation, rotation_format="angleaxis3", inverse_depth=False,
                   normalized_flow=False):
    if rotation_format != "angleaxis3":
        raise ValueError("only rotation_format='angleaxis3' is supported")
    return _ctx().flow_to_depth(flow, intrinsics, rotation, translation, inverse_depth, normalized_flow, method=1)


def warp2d(input, displacements, normalized=False, border_mode="clamp", border_value=0.0):
    if border_mode not in ("clamp", "value"):
        raise ValueError("border_mode must be 'clamp' or 'value'")
    return _ctx().warp2d(input, displacements, normalized, border_mode, border_value)


def leaky_relu(input, leak=0.1):
    return _ctx().leaky_relu(input, leak)


def replace_nonfinite(input, value=0.0):
    return _ctx().replace_nonfinite(input, value)


def scale_invariant_gradient(input, deltas=(1,), weights=(1.0,), epsilon=0.001):
    return _ctx().scale_invariant_gradient(input, deltas, weights, epsilon)


def median3x3_downsample(input):
    return _ctx().median3x3_downsample(input)


def depth_to_normals(depth, intrinsics, inverse_depth=False):
    """lmbspecialops.depth_to_normals as v2/losses.py:336-337 calls it.  UNVERIFIED against lmbspecialops (its sources are not in the
    reference tree): pixel centres at +0.5, the one-sided difference with the smaller magnitude, normals pointing towards the
    camera and NaN on the one-pixel border are this implementation's reading of the op's documentation (oracle/demon_oracle.c
    states the same); pinned only by an analytic plane test (tests/test_pins.py)."""
    return _ctx().depth_to_normals(depth, intrinsics, inverse_depth)
