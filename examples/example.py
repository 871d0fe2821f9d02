#!/usr/bin/env python3
"""DeMoN on an image pair with the MI355X-native path -- the flow of the reference's examples/example.py
(prepare inputs :15-42, bootstrap + 3 x iterative + refinement :87-99) on `depthmotionnet.networks_original`.

  python examples/example.py IMG1 IMG2 [--weights weights/demon_original | weights.npz | --synthetic] [--out result.npz]

The reference script itself also runs unmodified against this repo (see python/tf_stub/tensorflow/__init__.py).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "python"))
sys.path.insert(0, ROOT)

import demon_amd  # noqa: E402
from demon_amd import weights as W  # noqa: E402
from depthmotionnet.networks_original import BootstrapNet, IterativeNet, RefinementNet  # noqa: E402
from depthmotionnet.helpers import angleaxis_to_rotation_matrix  # noqa: E402


from demon_amd.preprocess import prepare_input_data  # noqa: E402  (reference examples/example.py:15-42)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("img1")
    ap.add_argument("img2")
    ap.add_argument("--weights", default=os.path.join(ROOT, "weights", "demon_original"))
    ap.add_argument("--synthetic", action="store_true", help="random weights (plumbing check without the checkpoint)")
    ap.add_argument("--data-format", default="channels_first", choices=["channels_first", "channels_last"])
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from PIL import Image

    if args.synthetic:
        w = W.synthetic_weights(seed=1)
    elif args.weights.endswith(".npz"):
        w = W.load_npz(args.weights)
    else:
        from demon_amd.tf_checkpoint import load_tf_checkpoint
        w = load_tf_checkpoint(args.weights, list(W.variable_shapes()))
    demon_amd.set_default_weights(w)

    data = prepare_input_data(Image.open(args.img1).convert("RGB"), Image.open(args.img2).convert("RGB"), args.data_format)
    bootstrap_net = BootstrapNet(None, args.data_format)
    iterative_net = IterativeNet(None, args.data_format)
    refine_net = RefinementNet(None, args.data_format)

    result = bootstrap_net.eval(data["image_pair"], data["image2_2"])
    for _ in range(3):
        result = iterative_net.eval(data["image_pair"], data["image2_2"], result["predict_depth2"], result["predict_normal2"],
                                    result["predict_rotation"], result["predict_translation"])
    rotation, translation = result["predict_rotation"], result["predict_translation"]
    depth0 = refine_net.eval(data["image1"], result["predict_depth2"])["predict_depth0"]

    print("rotation (angle axis):", rotation[0])
    print("rotation matrix:\n", angleaxis_to_rotation_matrix(rotation[0]))
    print("translation:", translation[0])
    print("inverse depth 192x256: min %.4f median %.4f max %.4f" % (depth0.min(), np.median(depth0), depth0.max()))
    if args.out:
        np.savez(args.out, predict_depth0=depth0, rotation=rotation, translation=translation)


if __name__ == "__main__":
    main()
