"""Generates tests/golden/sculpture_inputs.npz by RUNNING THE REFERENCE'S OWN prepare_input_data in this container.

  code   : /root/reference/examples/example.py:15-42 -- the function's source is cut out of the file with `ast` and exec'd
           unchanged (the rest of the script imports tensorflow and cannot run here)
  inputs : /root/reference/examples/sculpture1.png, sculpture2.png (the fixtures the reference ships)
  outputs: the decoded images (uint8, so that the tests need no PNG files), and for both data formats under two resize
           filters the function's image2_2 array in full plus the sha256 of the bytes of each of its three arrays (the
           256x192 arrays are 300 KB each; a hash pins them bit for bit):
             *_nearest : Image.resize defaulting to NEAREST, the behaviour of the Pillow 2.0.0 the reference pins
                         (Dockerfile:15; the default became BICUBIC in Pillow 7.0) -- obtained by wrapping Image.Image.resize
                         so that a call WITHOUT a filter argument gets NEAREST; the reference code itself is not edited
             *_pil     : the Pillow installed here (version stored), i.e. what the unmodified script computes today

Run:  python tests/golden/make_golden_inputs.py      (needs /root/reference; the GPU box only uses the .npz)
"""
import ast
import hashlib
import os

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def reference_function():
    path = os.path.join(REF, "examples", "example.py")
    src = open(path).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_input_data")
    code = ast.get_source_segment(src, node)
    ns = {"np": np}
    exec(compile(code, path, "exec"), ns)
    return ns["prepare_input_data"]


def big_pair(u8):
    """512x384 test image from a 256x192 one: pixel doubling plus a deterministic pattern (numpy integer arithmetic only)"""
    big = np.repeat(np.repeat(u8.astype(np.int32), 2, axis=0), 2, axis=1)
    y, x = np.mgrid[0:big.shape[0], 0:big.shape[1]]
    big = big + ((3 * x + 5 * y) % 17)[:, :, None] - 8
    return np.clip(big, 0, 255).astype(np.uint8)


def main():
    fn = reference_function()
    ex = os.path.join(REF, "examples")
    img1 = Image.open(os.path.join(ex, "sculpture1.png"))
    img2 = Image.open(os.path.join(ex, "sculpture2.png"))
    out = {"image1_u8": np.asarray(img1), "image2_u8": np.asarray(img2), "pillow_version": np.array(PIL.__version__)}
    # a larger pair so that the first two resize calls of the function are exercised as well; built with numpy only
    # (big_pair below), so the tests regenerate it exactly without storing it
    big1, big2 = (Image.fromarray(big_pair(np.asarray(im))) for im in (img1, img2))
    orig_resize = Image.Image.resize

    def nearest_default(self, size, resample=None, *args, **kwargs):
        return orig_resize(self, size, Image.NEAREST if resample is None else resample, *args, **kwargs)

    for tag in ("pil", "nearest"):
        if tag == "nearest":
            Image.Image.resize = nearest_default
        try:
            for fmt in ("channels_first", "channels_last"):
                for name, (a, b) in (("", (img1, img2)), ("big_", (big1, big2))):
                    r = fn(a, b, fmt)
                    for k, v in r.items():
                        key = "%s%s_%s_%s" % (name, k, fmt, tag)
                        assert v.dtype == np.float32
                        out["sha256_" + key] = np.array(hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest())
                        out["shape_" + key] = np.array(v.shape)
                        if k == "image2_2":
                            out[key] = v
        finally:
            Image.Image.resize = orig_resize
    path = os.path.join(HERE, "sculpture_inputs.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "entries")


if __name__ == "__main__":
    main()
