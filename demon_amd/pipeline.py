"""Host-resident batches through a lane group (demon_amd/lanes.py: several DemonContexts on one GPU) so that PCIe copies overlap the
kernels and several batches are in flight.  Lane k owns stream k: upload(batch i) -> forward graph -> download(batch i) are enqueued
asynchronously on it, batches go round the lanes, and the host only waits for a lane when it needs it again.  The reference's
counterpart is the prediction loop of examples/evaluation.py:225-256, which feeds one pair at a time through session.run.

Page-locking is the expensive part of the host side (hipHostRegister walks and pins every page: milliseconds per call for the
39 MB of one batch of inputs), so it is done ONCE per buffer, not once per run: `Pipeline.buffers(B)` hands out a set of
page-locked input / output arrays for B pairs that the caller fills, runs and reads any number of times (`HostBuffers`), and
`run()` on ordinary numpy arrays pins them for the duration of the call only (convenient, slower).
"""
import ctypes
import time

import numpy as np

from .engine import DemonContext, DemonError, DemonOutputs, _fp
from .lanes import LaneGroup


class _Pinned:
    """page-locks a numpy array for the lifetime of the object (hipHostRegister through the C ABI)"""

    def __init__(self, lib, arr):
        self.lib, self.ptr = lib, arr.ctypes.data
        self.ok = lib.demon_host_register(ctypes.c_void_p(self.ptr), arr.nbytes) == 0

    def release(self):
        if self.ok:
            self.lib.demon_host_unregister(ctypes.c_void_p(self.ptr))
            self.ok = False


class HostBuffers:
    """Page-locked host arrays for B pairs: `image_pair`, `image2_2` (inputs, filled by the caller) and `out[key]` (outputs)."""

    def __init__(self, lib, shapes, B, H, W):
        self.B = B
        self.image_pair = np.zeros((B, 6, H, W), np.float32)
        self.image2_2 = np.zeros((B, 3, H // 4, W // 4), np.float32)
        self.out = {k: np.zeros((B,) + s, np.float32) for k, s in shapes.items()}
        self._pins = [_Pinned(lib, a) for a in [self.image_pair, self.image2_2] + list(self.out.values())]
        self.pinned = all(p.ok for p in self._pins)

    def release(self):
        for p in self._pins:
            p.release()
        self._pins = []


class Pipeline:
    def __init__(self, weights, batch=32, height=192, width=256, device=0, version=1, contexts=3, calibrate=False):
        """contexts = lanes; calibrate=True: create `contexts` lanes, measure 2 .. contexts lanes on zero inputs and keep the best
        count (LaneGroup.calibrate; at least 2 lanes stay, so that copies still overlap kernels)"""
        self.batch, self.H, self.W = batch, height, width
        self.lanes = LaneGroup(weights, contexts, batch, height, width, device, version)
        self.ctxs = self.lanes.ctxs
        self.lane_rates = None
        if calibrate and contexts > 2:
            zp = np.zeros((batch, 6, height, width), np.float32)
            z2 = np.zeros((batch, 3, height // 4, width // 4), np.float32)
            self.lanes.upload_inputs([(zp, z2)] * len(self.ctxs))
            self.lane_rates = self.lanes.calibrate(batch, candidates=range(2, contexts + 1))
        c0 = self.ctxs[0]
        self.shapes = {
            "predict_flow5": (2, c0.h5, c0.w5), "predict_conf5": (2, c0.h5, c0.w5), "predict_flow2": (2, c0.h2, c0.w2),
            "predict_conf2": (2, c0.h2, c0.w2), "predict_depth2": (1, c0.h2, c0.w2), "predict_normal2": (3, c0.h2, c0.w2),
            "predict_rotation": (3,), "predict_translation": (3,), "predict_scale": (1,), "predict_depth0": (1, self.H, self.W),
        }

    def close(self):
        self.lanes.close()
        self.ctxs = []

    def buffers(self, B):
        """page-locked input / output arrays for B pairs (B a multiple of the batch size); release() them when done"""
        if B % self.batch:
            raise DemonError("B must be a multiple of the batch size %d" % self.batch)
        return HostBuffers(self.ctxs[0].lib, self.shapes, B, self.H, self.W)

    def run_buffers(self, hb, iterations=3):
        """every pair of `hb` through the pipeline: hb.image_pair / hb.image2_2 -> hb.out[...]; returns when everything has landed"""
        n, lib = self.batch, self.ctxs[0].lib
        for i in range(hb.B // n):
            c = self.ctxs[i % len(self.ctxs)]
            if i >= len(self.ctxs):
                c.synchronize()          # its previous batch (inputs consumed, outputs written)
            sl = slice(i * n, (i + 1) * n)
            c._check(lib.demon_upload_inputs_async(c.h, n, _fp(hb.image_pair[sl]), _fp(hb.image2_2[sl])))
            c.run_full(n, iterations)
            o = DemonOutputs(**{k: _fp(hb.out[k][sl]) for k in DemonContext.OUTPUT_KEYS})
            c._check(lib.demon_download_outputs_async(c.h, n, ctypes.byref(o), _fp(hb.out["predict_depth0"][sl])))
        for c in self.ctxs:
            c.synchronize()
        return hb.out

    def throughput(self, hb, iterations=3, repeats=3):
        """host-to-host pairs/s of run_buffers (one untimed pass first); {"pairs_per_s", "ms_per_batch", "pinned"}"""
        self.run_buffers(hb, iterations)
        t0 = time.perf_counter()
        for _ in range(repeats):
            self.run_buffers(hb, iterations)
        dt = (time.perf_counter() - t0) / repeats
        return {"pairs_per_s": hb.B / dt, "ms_per_batch": 1e3 * dt * self.batch / hb.B, "pinned": bool(hb.pinned),
                "pairs_per_pass": hb.B, "contexts": len(self.ctxs)}

    def run(self, image_pair, image2_2, iterations=3):
        """image_pair [B,6,H,W], image2_2 [B,3,H/4,W/4] float32 host arrays, B a multiple of the batch size.
        Returns dict of host arrays (the keys of DemonContext.full) for all B pairs.  The caller's arrays are page-locked for the
        duration of the call; use buffers() + run_buffers() to pay for that once."""
        image_pair = np.ascontiguousarray(image_pair, np.float32)
        image2_2 = np.ascontiguousarray(image2_2, np.float32)
        B, n = image_pair.shape[0], self.batch
        if B % n or image_pair.shape[1:] != (6, self.H, self.W) or image2_2.shape != (B, 3, self.H // 4, self.W // 4):
            raise DemonError("inputs must be [k*batch,6,H,W] and [k*batch,3,H/4,W/4]")
        hb = HostBuffers.__new__(HostBuffers)
        hb.B, hb.image_pair, hb.image2_2 = B, image_pair, image2_2
        hb.out = {k: np.empty((B,) + s, np.float32) for k, s in self.shapes.items()}
        lib = self.ctxs[0].lib
        hb._pins = [_Pinned(lib, a) for a in [image_pair, image2_2] + list(hb.out.values())]
        hb.pinned = all(p.ok for p in hb._pins)
        try:
            return self.run_buffers(hb, iterations)
        finally:
            hb.release()
