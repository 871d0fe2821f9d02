"""Generates tests/golden/wiring_losses.npz by RUNNING THE REFERENCE'S OWN v2/losses.py in this container.

  code   : /root/reference/python/depthmotionnet/v2/losses.py:23-104, :312-374 and v2/helpers.py:94-104, imported unmodified on the
           TensorFlow-1.4 graph emulator oracle/tf1/ (`lmbspecialops` = oracle/ops_ref.py: the ops' own semantics stay unpinned)
  inputs : seeded inverse-depth map [2,1,64,96] with invalid (NaN) pixels, camera motions, seeded prediction / ground-truth pairs
  outputs: prepare_ground_truth_tensors (the median pyramid, depth -> flow at three levels, normals, the scale invariant gradient
           images with their one-call-per-delta concatenation), pointwise_l2_loss (both data formats), scale_invariant_gradient_loss,
           l1_loss, compute_confidence_map

Run:  python tests/golden/make_golden_losses.py      (needs /root/reference).  tests/test_wiring_goldens.py holds demon_amd/losses.py
to the file: on the CPU with the oracle's ops behind it, on the GPU with the HIP ops.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DEMON_REFERENCE", "/root/reference")
INTRINSICS = (0.89115971, 1.18821287, 0.5, 0.5)


def inputs():
    rs = np.random.RandomState(17)
    n, h, w = 2, 64, 96
    depth = (0.3 + rs.random_sample((n, 1, h, w))).astype(np.float32)
    depth[0, 0, 5:9, 10:20] = np.nan
    depth[1, 0, 40, 50] = np.nan
    rot = (0.05 * rs.standard_normal((n, 3))).astype(np.float32)
    tr = rs.standard_normal((n, 3)).astype(np.float32)
    tr /= np.linalg.norm(tr, axis=1, keepdims=True)
    K = np.tile(np.array([INTRINSICS], np.float32), (n, 1))
    pred = rs.standard_normal((n, 10, 16, 24)).astype(np.float32)
    gt = (pred + 0.3 * rs.standard_normal(pred.shape)).astype(np.float32)
    gt[0, :, 3, 4] = np.nan
    flow_p = (0.05 * rs.standard_normal((n, 2, 16, 24))).astype(np.float32)
    flow_g = (0.05 * rs.standard_normal((n, 2, 16, 24))).astype(np.float32)
    return {"depth": depth, "rotation": rot, "translation": tr, "intrinsics": K, "pred": pred, "gt": gt, "flow_p": flow_p, "flow_g": flow_g}


def main():
    for p in (os.path.join(REF, "python"), ROOT, os.path.join(ROOT, "oracle", "tf1")):
        sys.path.insert(0, p)
    import tensorflow as tf
    assert tf.EMULATED
    from depthmotionnet.v2 import losses as L
    x = inputs()
    out = {}
    with tf.Graph().as_default(), tf.Session() as s:
        c = dict((k, tf.constant(v)) for k, v in x.items())
        g = L.prepare_ground_truth_tensors(c["depth"], c["rotation"], c["translation"], c["intrinsics"])
        fetch = dict(("gt/" + k, v) for k, v in g.items())
        fetch["pointwise_l2_loss/nchw"] = L.pointwise_l2_loss(c["pred"], c["gt"], 0.01)
        fetch["pointwise_l2_loss/nhwc"] = L.pointwise_l2_loss(tf.transpose(c["pred"], [0, 2, 3, 1]), tf.transpose(c["gt"], [0, 2, 3, 1]), 0.01, data_format="NHWC")
        fetch["scale_invariant_gradient_loss"] = L.scale_invariant_gradient_loss(c["pred"], c["gt"], 0.01)
        fetch["l1_loss"] = L.l1_loss(c["flow_p"], 0.001)
        fetch["compute_confidence_map"] = L.compute_confidence_map(c["flow_p"], c["flow_g"], scale=2)
        fetch["scale_invariant_gradient"] = L.scale_invariant_gradient(c["flow_p"], deltas=[1, 2, 4], weights=[1, 0.5, 0.25], epsilon=0.001)
        out = s.run(fetch)
    np.savez_compressed(os.path.join(HERE, "wiring_losses.npz"), backend=np.array("reference v2/losses.py on oracle/tf1"),
                        **dict((k, np.asarray(v, np.float32)) for k, v in out.items()))
    print("wiring_losses.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "wiring_losses.npz")), "bytes")
    for k in sorted(out):
        print("  %-36s %s" % (k, np.asarray(out[k]).shape))


if __name__ == "__main__":
    main()
