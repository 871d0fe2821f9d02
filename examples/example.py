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


def prepare_input_data(img1, img2, data_format):
    """PIL images -> network inputs in [-0.5, 0.5] (reference examples/example.py:15-42).  The reference relies on
    PIL's default resize filter, which changed between Pillow versions (SURVEY hazard H3): NEAREST is used here, the
    default of the Pillow 2.0.0 the reference pins (Dockerfile:15)."""
    from PIL import Image
    if img1.size != (256, 192):
        img1 = img1.resize((256, 192), Image.NEAREST)
    if img2.size != (256, 192):
        img2 = img2.resize((256, 192), Image.NEAREST)
    img2_2 = img2.resize((64, 48), Image.NEAREST)
    arrs = [np.asarray(im.convert("RGB"), dtype=np.float32) / 255 - 0.5 for im in (img1, img2, img2_2)]
    if data_format == "channels_first":
        arrs = [a.transpose(2, 0, 1) for a in arrs]
        pair = np.concatenate(arrs[:2], axis=0)
    else:
        pair = np.concatenate(arrs[:2], axis=-1)
    return {"image_pair": pair[np.newaxis], "image1": arrs[0][np.newaxis], "image2_2": arrs[2][np.newaxis]}


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

    data = prepare_input_data(Image.open(args.img1), Image.open(args.img2), args.data_format)
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
