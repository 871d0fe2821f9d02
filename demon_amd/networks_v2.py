"""Mirror of `depthmotionnet.v2.networks` (reference python/depthmotionnet/v2/networks.py) on top of
libdemon_hip.so: the retrained model of v2/blocks.py behind the same three classes the v2 driver uses
(examples/example_v2.py:81-105).  Constructors take only `session` and the placeholders are
channels_first with batch 1, as in the reference (v2/networks.py:21-31, :81-96, :182-196); `batch_size`
is an extension.  Weights: an object with a `demon_weights` attribute, or demon_amd.set_default_weights()
with the v2 variable set (demon_amd.weights.variable_shapes(version=2)).
"""
from . import runtime
from .networks_original import _H, _W, _Net, _to_nchw

__all__ = ["BootstrapNet", "IterativeNet", "RefinementNet"]


class _NetV2(_Net):
    _version = 2

    def __init__(self, session, batch_size=1):
        super().__init__(session, "channels_first", batch_size)


class BootstrapNet(_NetV2):
    """reference v2/networks.py:20-76"""

    def eval(self, image_pair, image2_2):
        image_pair = self._check(image_pair, self._shape(6, _H, _W), "placeholder_image_pair")
        image2_2 = self._check(image2_2, self._shape(3, _H // 4, _W // 4), "placeholder_image2_2")
        out = self._outputs(self._ctx.bootstrap(_to_nchw(image_pair, "channels_first"), _to_nchw(image2_2, "channels_first")))
        runtime.note_eval(self, {"image_pair": image_pair, "image2_2": image2_2}, out)
        return out


class IterativeNet(_NetV2):
    """reference v2/networks.py:80-176"""

    def eval(self, image_pair, image2_2, depth2, normal2, rotation, translation):
        df = "channels_first"
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


class RefinementNet(_NetV2):
    """reference v2/networks.py:181-232 (normal2 is fed but not used by the block, v2/blocks.py:499-527)"""

    def eval(self, image1, depth2, normal2):
        image1 = self._check(image1, self._shape(3, _H, _W), "placeholder_image1")
        depth2 = self._check(depth2, self._shape(1, _H // 4, _W // 4), "placeholder_depth2")
        self._check(normal2, self._shape(3, _H // 4, _W // 4), "placeholder_normal2")
        r = self._ctx.refine(_to_nchw(image1, "channels_first"), _to_nchw(depth2, "channels_first"))
        out = {"predict_depth0": r["predict_depth0"], "predict_normal0": r["predict_normal0"]}
        runtime.note_eval(self, {"image1": image1}, out)
        return out
