"""Host-side input preparation of the drivers: reference examples/example.py:15-42 (`prepare_input_data`).

Pure PIL / numpy, like the reference.  The reference calls `img.resize(size)` WITHOUT a filter, so its result depends on the
installed Pillow: NEAREST up to Pillow 6 -- including the 2.0.0 the reference pins (Dockerfile:15) -- and BICUBIC from 7.0 on
(hazard 1).  `resample` makes the choice explicit:
  "reference" (default)  NEAREST, what the reference's own environment computes
  "pil"                  no filter argument, i.e. whatever the installed Pillow defaults to (what the unmodified script does today)
  any PIL filter constant
tests/test_preprocess.py holds both modes to arrays produced by the reference function itself (tests/golden/make_golden_inputs.py).
"""
import numpy as np


def _resize(img, size, resample):
    from PIL import Image
    if resample == "pil":
        return img.resize(size)
    return img.resize(size, Image.NEAREST if resample == "reference" else resample)


def prepare_input_data(img1, img2, data_format, resample="reference"):
    """PIL images -> {'image_pair' [1,6,192,256], 'image1' [1,3,192,256], 'image2_2' [1,3,48,64]} in [-0.5, 0.5]
    (channels_last: [1,192,256,6], [1,192,256,3], [1,48,64,3]).  Same keys, shapes, dtype and values as the reference."""
    if data_format not in ("channels_first", "channels_last"):
        raise ValueError("data_format must be 'channels_first' or 'channels_last'")
    # scale images if necessary (:18-22); the quarter-size second image is made from the (resized) second image
    if img1.size[0] != 256 or img1.size[1] != 192:
        img1 = _resize(img1, (256, 192), resample)
    if img2.size[0] != 256 or img2.size[1] != 192:
        img2 = _resize(img2, (256, 192), resample)
    img2_2 = _resize(img2, (64, 48), resample)
    # [0, 255] -> [-0.5, 0.5] (:25-27): float32 division, then float32 subtraction, as numpy does for the reference
    arrs = [np.array(im).astype(np.float32) / 255 - 0.5 for im in (img1, img2, img2_2)]
    if data_format == "channels_first":
        arrs = [a.transpose([2, 0, 1]) for a in arrs]
        pair = np.concatenate((arrs[0], arrs[1]), axis=0)
    else:
        pair = np.concatenate((arrs[0], arrs[1]), axis=-1)
    return {"image_pair": pair[np.newaxis, :], "image1": arrs[0][np.newaxis, :], "image2_2": arrs[2][np.newaxis, :]}
