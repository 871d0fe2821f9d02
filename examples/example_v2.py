#!/usr/bin/env python3
"""The v2 DeMoN model on an image pair with the MI355X-native path -- the flow of the reference's examples/example_v2.py
(--checkpoint :15-17, prepare inputs :30-47, bootstrap + 3 x iterative + refinement :93-105) on `depthmotionnet.v2.networks`.

  python examples/example_v2.py IMG1 IMG2 --checkpoint PREFIX | --synthetic  [--out result.npz]

The reference script itself also runs unmodified against this repo (see python/tf_stub/tensorflow/__init__.py).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "python"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import demon_amd  # noqa: E402
from demon_amd import weights as W  # noqa: E402
from depthmotionnet.v2.networks import BootstrapNet, IterativeNet, RefinementNet  # noqa: E402
from example import prepare_input_data  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("img1")
    ap.add_argument("img2")
    ap.add_argument("--checkpoint", default="", help="TF checkpoint prefix of a trained v2 model (or a .npz of its variables)")
    ap.add_argument("--synthetic", action="store_true", help="random weights (plumbing check without a checkpoint)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from PIL import Image

    if args.synthetic:
        w = W.synthetic_weights(seed=1, version=2)
    elif args.checkpoint.endswith(".npz"):
        w = W.load_npz(args.checkpoint)
    elif args.checkpoint:
        from demon_amd.tf_checkpoint import load_tf_checkpoint
        w = load_tf_checkpoint(args.checkpoint, list(W.variable_shapes(version=2)))
    else:
        ap.error("--checkpoint or --synthetic is required (the reference ships no v2 weights)")
    demon_amd.set_default_weights(w)

    data = prepare_input_data(Image.open(args.img1).convert("RGB"), Image.open(args.img2).convert("RGB"), "channels_first")
    bootstrap_net, iterative_net, refine_net = BootstrapNet(None), IterativeNet(None), RefinementNet(None)
    result = bootstrap_net.eval(data["image_pair"], data["image2_2"])
    for _ in range(3):
        result = iterative_net.eval(data["image_pair"], data["image2_2"], result["predict_depth2"], result["predict_normal2"],
                                    result["predict_rotation"], result["predict_translation"])
    rotation, translation = result["predict_rotation"], result["predict_translation"]
    result = refine_net.eval(data["image1"], result["predict_depth2"], result["predict_normal2"])
    depth0, normal0 = result["predict_depth0"], result["predict_normal0"]

    print("rotation (angle axis):", rotation[0])
    print("translation:", translation[0])
    print("inverse depth 192x256: min %.4f median %.4f max %.4f" % (depth0.min(), np.median(depth0), depth0.max()))
    print("normal 192x256: mean", normal0.mean(axis=(0, 2, 3)))
    if args.out:
        np.savez(args.out, predict_depth0=depth0, predict_normal0=normal0, rotation=rotation, translation=translation)


if __name__ == "__main__":
    main()
