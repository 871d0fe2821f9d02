"""Mirror of `depthmotionnet.networks_original` (reference python/depthmotionnet/networks_original.py)
on top of libdemon_hip.so: same class names, constructor arguments, eval() arguments, returned dict
keys and array shapes for both data formats, so examples/example.py:75-99 runs unchanged.

There is no TensorFlow graph here: each eval() is one call into the C ABI (one hipGraph launch on the
GPU).  `session` is accepted for signature compatibility; an object with a `demon_weights` attribute
(see demon_amd.tf_stub) supplies the weights, otherwise demon_amd.set_default_weights() does.
"""
import numpy as np

from . import runtime

__all__ = ["BootstrapNet", "IterativeNet", "RefinementNet"]

_H, _W = 192, 256  # fixed by the reference's placeholders (networks_original.py:38-42)


def _to_nchw(a, data_format):
    a = np.asarray(a, dtype=np.float32)
    if data_format == "channels_last" and a.ndim == 4:
        a = a.transpose(0, 3, 1, 2)
    return np.ascontiguousarray(a)


def _from_nchw(a, data_format):
    if data_format == "channels_last" and a.ndim == 4:
        return np.ascontiguousarray(a.transpose(0, 2, 3, 1))
    return a


class _Net:
    _version = 1

    def __init__(self, session, data_format="channels_first", batch_size=1):
        if data_format not in ("channels_first", "channels_last"):
            raise ValueError("data_format must be 'channels_first' or 'channels_last'")
        self.session = session
        self.data_format = data_format
        self.batch_size = batch_size
        # a session that carries its own variables (python/tf_stub) owns its context; Saver.restore fills it later
        self._ctx = runtime.get_context(batch_size, _H, _W, version=self._version, session=session)

    def _shape(self, c, h, w):
        n = self.batch_size
        return (n, c, h, w) if self.data_format == "channels_first" else (n, h, w, c)

    def _check(self, arr, shape, name):
        arr = np.asarray(arr)
        if tuple(arr.shape) != tuple(shape):
            # TF: "Cannot feed value of shape ... for Tensor ..., which has shape ..."
            raise ValueError("Cannot feed value of shape %s for %s, which has shape %s"
                             % (tuple(arr.shape), name, tuple(shape)))
        return arr

    def _outputs(self, r):
        keys = ("predict_flow5", "predict_flow2", "predict_depth2", "predict_normal2", "predict_rotation",
                "predict_translation")
        return {k: _from_nchw(r[k], self.data_format) for k in keys}


class BootstrapNet(_Net):
    """reference networks_original.py:22-88"""

    def eval(self, image_pair, image2_2):
        image_pair = self._check(image_pair, self._shape(6, _H, _W), "placeholder_image_pair")
        image2_2 = self._check(image2_2, self._shape(3, _H // 4, _W // 4), "placeholder_image2_2")
        r = self._ctx.bootstrap(_to_nchw(image_pair, self.data_format), _to_nchw(image2_2, self.data_format))
        out = self._outputs(r)
        runtime.note_eval(self, {"image_pair": image_pair, "image2_2": image2_2}, out)
        return out


class IterativeNet(_Net):
    """reference networks_original.py:92-198"""

    def __init__(self, session, data_format="channels_first", batch_size=1):
        super().__init__(session, data_format, batch_size)
        self.intrinsics = np.broadcast_to(np.array([[0.89115971, 1.18821287, 0.5, 0.5]], np.float32), (batch_size, 4))

    def eval(self, image_pair, image2_2, depth2, normal2, rotation, translation):
        df = self.data_format
        image_pair = self._check(image_pair, self._shape(6, _H, _W), "placeholder_image_pair")
        image2_2 = self._check(image2_2, self._shape(3, _H // 4, _W // 4), "placeholder_image2_2")
        depth2 = self._check(depth2, self._shape(1, _H // 4, _W // 4), "placeholder_depth2")
        normal2 = self._check(normal2, self._shape(3, _H // 4, _W // 4), "placeholder_normal2")
        rotation = self._check(rotation, (self.batch_size, 3), "placeholder_rotation")
        translation = self._check(translation, (self.batch_size, 3), "placeholder_translation")
        r = self._ctx.iterative(_to_nchw(image_pair, df), _to_nchw(image2_2, df), _to_nchw(depth2, df),
                                _to_nchw(normal2, df), _to_nchw(rotation, df), _to_nchw(translation, df))
        out = self._outputs(r)
        runtime.note_eval(self, {}, out)
        return out


class RefinementNet(_Net):
    """reference networks_original.py:202-255"""

    def eval(self, image1, depth2):
        image1 = self._check(image1, self._shape(3, _H, _W), "placeholder_image1")
        depth2 = self._check(depth2, self._shape(1, _H // 4, _W // 4), "placeholder_depth2")
        r = self._ctx.refine(_to_nchw(image1, self.data_format), _to_nchw(depth2, self.data_format))
        out = {"predict_depth0": _from_nchw(r["predict_depth0"], self.data_format)}
        runtime.note_eval(self, {"image1": image1}, out)
        return out
