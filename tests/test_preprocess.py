"""prepare_input_data against arrays produced by the REFERENCE function itself
(examples/example.py:15-42, exec'd by tests/golden/make_golden_inputs.py in the build container)."""
import hashlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
GOLDEN = os.path.join(HERE, "golden", "sculpture_inputs.npz")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("fmt", ["channels_first", "channels_last"])
@pytest.mark.parametrize("pair", ["", "big_"])
def test_prepare_input_data_matches_reference_function(fmt, pair):
    import PIL
    from PIL import Image
    from demon_amd.preprocess import prepare_input_data
    from make_golden_inputs import big_pair
    g = np.load(GOLDEN)
    u1, u2 = g["image1_u8"], g["image2_u8"]
    if pair:
        u1, u2 = big_pair(u1), big_pair(u2)   # 512x384: exercises the two full-size resize calls (:18-21)
    img1, img2 = Image.fromarray(u1), Image.fromarray(u2)
    modes = [("reference", "nearest")]        # Pillow 2.0.0 behaviour the reference pins (Dockerfile:15)
    if str(g["pillow_version"]) == PIL.__version__:
        modes.append(("pil", "pil"))          # the installed Pillow's default filter: only comparable on the same Pillow
    for resample, tag in modes:
        r = prepare_input_data(img1, img2, fmt, resample=resample)
        assert sorted(r) == ["image1", "image2_2", "image_pair"]
        for k, v in r.items():
            key = "%s%s_%s_%s" % (pair, k, fmt, tag)
            assert v.dtype == np.float32 and tuple(v.shape) == tuple(g["shape_" + key]), key
            if k == "image2_2":
                np.testing.assert_array_equal(v, g[key], err_msg=key)
            assert _sha(v) == str(g["sha256_" + key]), key   # bit for bit


def test_prepare_input_data_layout_and_range():
    """image_pair = [img1 RGB, img2 RGB] (example.py:33), values (u8 / 255 - 0.5), image1 is the first half of the pair"""
    from PIL import Image
    from demon_amd.preprocess import prepare_input_data
    g = np.load(GOLDEN)
    r = prepare_input_data(Image.fromarray(g["image1_u8"]), Image.fromarray(g["image2_u8"]), "channels_first")
    assert r["image_pair"].shape == (1, 6, 192, 256) and r["image1"].shape == (1, 3, 192, 256) and r["image2_2"].shape == (1, 3, 48, 64)
    np.testing.assert_array_equal(r["image_pair"][:, :3], r["image1"])
    np.testing.assert_array_equal(r["image1"][0], (g["image1_u8"].astype(np.float32) / 255 - 0.5).transpose(2, 0, 1))
    np.testing.assert_array_equal(r["image_pair"][0, 3:], (g["image2_u8"].astype(np.float32) / 255 - 0.5).transpose(2, 0, 1))
    assert r["image_pair"].min() >= -0.5 and r["image_pair"].max() <= 0.5
    # NEAREST 4x reduction picks the pixel whose centre maps to the output centre: index 4*i + 2
    np.testing.assert_array_equal(r["image2_2"][0], r["image_pair"][0, 3:, 2::4, 2::4])
    with pytest.raises(ValueError):
        prepare_input_data(Image.fromarray(g["image1_u8"]), Image.fromarray(g["image2_u8"]), "NCHW")
