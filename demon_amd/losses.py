"""Forward values of the ground-truth preparation and loss helpers of the reference's v2 training code
(python/depthmotionnet/v2/losses.py, v2/helpers.py:94-104) on numpy arrays, every op executed by the HIP kernels of
libdemon_hip.so through `demon_amd.sops`.  Same function names, argument names and tensor
layouts (NCHW) as the reference; there is no autograd here -- training itself is out of scope -- these are
the forward values: what `prepare_ground_truth_tensors` feeds the losses, and what the losses evaluate to.
"""
import numpy as np

from . import sops


def recursive_median_downsample(inp, iterations):
    """v2/helpers.py:94-104"""
    result = []
    for _ in range(iterations):
        result.append(sops.median3x3_downsample(inp if not result else result[-1]))
    return tuple(result)


def pointwise_l2_loss(inp, gt, epsilon, data_format="NCHW"):
    """v2/losses.py:33-54"""
    if data_format != "NCHW":
        inp, gt = np.moveaxis(np.asarray(inp), 3, 1), np.moveaxis(np.asarray(gt), 3, 1)
    return sops._ctx().pointwise_l2_loss(inp, gt, epsilon)


def scale_invariant_gradient(inp, deltas, weights, epsilon=0.001):
    """v2/losses.py:57-79: one op call per delta, concatenated on axis 1.  The op folds channels into the batch, so
    [N,C,H,W] -> [N*C, 2*len(deltas), H, W] (flow2: [2N,10,H,W], as in the reference)"""
    assert len(deltas) == len(weights)
    return np.concatenate([sops.scale_invariant_gradient(inp, deltas=[d], weights=[w], epsilon=epsilon)
                           for d, w in zip(deltas, weights)], axis=1)


def scale_invariant_gradient_loss(inp, gt, epsilon):
    """v2/losses.py:83-104: sum over the (x, y) channel pairs of the pointwise l2 loss"""
    inp, gt = np.asarray(inp), np.asarray(gt)
    assert inp.shape[1] % 2 == 0 and inp.shape[1] == gt.shape[1]
    return sum(pointwise_l2_loss(inp[:, 2 * i:2 * i + 2], gt[:, 2 * i:2 * i + 2], epsilon) for i in range(inp.shape[1] // 2))


def compute_confidence_map(predicted_flow, gt_flow, scale=1):
    """v2/losses.py:359-374"""
    return np.exp(-scale * np.abs(np.asarray(predicted_flow, np.float32) - np.asarray(gt_flow, np.float32))).astype(np.float32)


def prepare_ground_truth_tensors(depth, rotation, translation, intrinsics):
    """v2/losses.py:312-356: lower-resolution ground truth (3x3 median pyramid), flow and normals from the inverse depth map
    and the camera motion, scale invariant gradient images.  depth [N,1,H,W] inverse depth."""
    depth1, depth2, depth3, depth4, depth5 = recursive_median_downsample(depth, 5)
    flow0 = sops.depth_to_flow(depth, intrinsics, rotation, translation, inverse_depth=True, normalize_flow=True)
    flow2 = sops.depth_to_flow(depth2, intrinsics, rotation, translation, inverse_depth=True, normalize_flow=True)
    flow5 = sops.depth_to_flow(depth5, intrinsics, rotation, translation, inverse_depth=True, normalize_flow=True)
    normal0 = sops.depth_to_normals(depth, intrinsics, inverse_depth=True)
    normal2 = sops.depth_to_normals(depth2, intrinsics, inverse_depth=True)
    sig_params = {"deltas": [1, 2, 4, 8, 16], "weights": [1, 1, 1, 1, 1], "epsilon": 0.001}
    return {
        "depth0": depth, "depth0_sig": scale_invariant_gradient(depth, **sig_params),
        "depth2": depth2, "depth2_sig": scale_invariant_gradient(depth2, **sig_params),
        "flow0": flow0, "flow2": flow2, "flow2_sig": scale_invariant_gradient(flow2, **sig_params), "flow5": flow5,
        "normal0": normal0, "normal2": normal2,
    }
