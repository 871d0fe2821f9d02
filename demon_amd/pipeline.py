"""Host-resident batches through two DemonContexts so that PCIe copies overlap the kernels (copy / compute overlap on separate
HIP streams).  Context k owns stream k: upload(batch i) -> forward graph -> download(batch i) are enqueued asynchronously on it,
batches alternate between the contexts, and the host only waits for a context when it needs it again.  The reference's
counterpart is the prediction loop of examples/evaluation.py:225-256, which feeds one pair at a time through session.run.
"""
import ctypes

import numpy as np

from .engine import DemonContext, DemonError, DemonOutputs, _fp


class _Pinned:
    """page-locks a numpy array for the lifetime of the object (hipHostRegister through the C ABI)"""

    def __init__(self, lib, arr):
        self.lib, self.ptr = lib, arr.ctypes.data
        self.ok = lib.demon_host_register(ctypes.c_void_p(self.ptr), arr.nbytes) == 0

    def release(self):
        if self.ok:
            self.lib.demon_host_unregister(ctypes.c_void_p(self.ptr))
            self.ok = False


class Pipeline:
    def __init__(self, weights, batch=32, height=192, width=256, device=0, version=1, contexts=2):
        self.batch, self.H, self.W = batch, height, width
        self.ctxs = []
        for _ in range(contexts):
            c = DemonContext(device, batch, height, width, version)
            c.set_weights(weights)
            c.load_tuned_plan(batch)
            self.ctxs.append(c)

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []

    def run(self, image_pair, image2_2, iterations=3):
        """image_pair [B,6,H,W], image2_2 [B,3,H/4,W/4] float32 host arrays, B a multiple of the batch size.
        Returns dict of host arrays (the keys of DemonContext.full) for all B pairs."""
        image_pair = np.ascontiguousarray(image_pair, np.float32)
        image2_2 = np.ascontiguousarray(image2_2, np.float32)
        B, n = image_pair.shape[0], self.batch
        if B % n or image_pair.shape[1:] != (6, self.H, self.W) or image2_2.shape != (B, 3, self.H // 4, self.W // 4):
            raise DemonError("inputs must be [k*batch,6,H,W] and [k*batch,3,H/4,W/4]")
        c0 = self.ctxs[0]
        out = {
            "predict_flow5": (2, c0.h5, c0.w5), "predict_conf5": (2, c0.h5, c0.w5), "predict_flow2": (2, c0.h2, c0.w2),
            "predict_conf2": (2, c0.h2, c0.w2), "predict_depth2": (1, c0.h2, c0.w2), "predict_normal2": (3, c0.h2, c0.w2),
            "predict_rotation": (3,), "predict_translation": (3,), "predict_scale": (1,), "predict_depth0": (1, self.H, self.W),
        }
        out = {k: np.empty((B,) + s, np.float32) for k, s in out.items()}
        lib = c0.lib
        pins = [_Pinned(lib, a) for a in [image_pair, image2_2] + list(out.values())]
        try:
            for i in range(B // n):
                c = self.ctxs[i % len(self.ctxs)]
                if i >= len(self.ctxs):
                    c.synchronize()          # its previous batch (inputs consumed, outputs written)
                sl = slice(i * n, (i + 1) * n)
                c._check(lib.demon_upload_inputs_async(c.h, n, _fp(image_pair[sl]), _fp(image2_2[sl])))
                c.run_full(n, iterations)
                o = DemonOutputs(**{k: _fp(out[k][sl]) for k in DemonContext.OUTPUT_KEYS})
                c._check(lib.demon_download_outputs_async(c.h, n, ctypes.byref(o), _fp(out["predict_depth0"][sl])))
            for c in self.ctxs:
                c.synchronize()
        finally:
            for p in pins:
                p.release()
        return out
