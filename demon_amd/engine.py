"""DemonContext -- Python owner of one demon_ctx (one GPU, one HIP stream)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import DemonOutputs, LaunchRecord, c_float_p, c_int64_p


class DemonError(ValueError):
    """Raised for every non-zero demon_status (the reference raises ValueError / InvalidArgumentError)."""


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _f32(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise DemonError("%s has shape %s, expected %s" % (name, tuple(a.shape), tuple(shape)))
    return a


class DemonContext:
    created_in_process = 0     # contexts ever created here: the HIP runtime is initialised once this is > 0 (demon_amd.lanes)
    OUTPUT_KEYS = ("predict_flow5", "predict_conf5", "predict_flow2", "predict_conf2", "predict_depth2",
                   "predict_normal2", "predict_rotation", "predict_translation", "predict_scale")

    def __init__(self, device=0, max_batch=1, height=192, width=256, version=1):
        """version 1: networks_original.py / blocks_original.py; version 2: v2/networks.py / v2/blocks.py"""
        if version not in (1, 2):
            raise DemonError("version must be 1 (original) or 2 (v2)")
        self.lib = _lib.load()
        self.h = ctypes.c_void_p()
        self.version = version
        create = self.lib.demon_create if version == 1 else self.lib.demon_create_v2
        rc = create(ctypes.byref(self.h), device, max_batch, height, width)
        if rc != 0:
            raise DemonError("demon_create failed (%d): %s" % (rc, self.lib.demon_last_error(None).decode()))
        DemonContext.created_in_process += 1
        self.device, self.max_batch, self.H, self.W = device, max_batch, height, width
        self.h2, self.w2, self.h5, self.w5 = height // 4, width // 4, height // 32, width // 32

    @classmethod
    def ops_only(cls, device=0):
        """demon_create_ops: a context with a stream and the split-K workspace only, for the demon_op_* entry points"""
        self = cls.__new__(cls)
        self.lib = _lib.load()
        self.h = ctypes.c_void_p()
        self.version = 0
        rc = self.lib.demon_create_ops(ctypes.byref(self.h), device)
        if rc != 0:
            raise DemonError("demon_create_ops failed (%d): %s" % (rc, self.lib.demon_last_error(None).decode()))
        DemonContext.created_in_process += 1
        self.device, self.max_batch, self.H, self.W = device, 0, 0, 0
        self.h2 = self.w2 = self.h5 = self.w5 = 0
        return self

    def close(self):
        if getattr(self, "h", None):
            self.lib.demon_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise DemonError("libdemon_hip error %d: %s" % (rc, self.lib.demon_last_error(self.h).decode()))

    # ---- weights ----------------------------------------------------------------------------------------
    def variables(self):
        """[(tf_name, shape)] in blob order."""
        out = []
        name = ctypes.create_string_buffer(128)
        dims = (ctypes.c_int64 * 4)()
        nd = ctypes.c_int()
        for i in range(self.lib.demon_num_variables(self.h)):
            self._check(self.lib.demon_variable_info(self.h, i, name, 128, dims, ctypes.byref(nd)))
            out.append((name.value.decode(), tuple(int(dims[k]) for k in range(nd.value))))
        return out

    def set_weights(self, weights):
        """weights: dict tf_name -> array in TF layout; every variable must be present."""
        variables = self.variables()
        checked = []
        for name, shape in variables:   # validate everything first: a failed call must not leave a half-loaded model
            if name not in weights:
                raise DemonError("missing variable %s" % name)
            checked.append((name, shape, _f32(weights[name], shape, name)))
        for name, shape, w in checked:
            dims = (ctypes.c_int64 * len(shape))(*shape)
            self._check(self.lib.demon_set_weight(self.h, name.encode(), _fp(w), dims, len(shape)))

    def blob_size(self):
        return int(self.lib.demon_weights_blob_size(self.h))

    def set_weights_blob(self, blob):
        blob = _f32(blob, (self.blob_size(),), "weight blob")
        self._check(self.lib.demon_set_weights_blob(self.h, _fp(blob), blob.size))

    def set_weights_blob_device(self, device_ptr, nfloats):
        self._check(self.lib.demon_set_weights_blob_device(self.h, ctypes.c_void_p(device_ptr), nfloats))

    def copy_weights_from(self, src):
        """demon_copy_weights_from: the packed weight slab of `src` device to device, then the epilogue of a broadcast receiver"""
        self._check(self.lib.demon_copy_weights_from(self.h, src.h))

    def slab_layout(self):
        return int(self.lib.demon_weights_slab_layout(self.h))

    def autotune(self, n):
        """measure every kernel variant per layer at batch n and keep the fastest (launch plans only)"""
        self._check(self.lib.demon_autotune(self.h, int(n)))

    def get_plan(self, n):
        """{layer name: [kind, tile, ksplit]} of the layers that have a tuned plan for batch n"""
        out = {}
        name = ctypes.create_string_buffer(128)
        kind, tile, ks = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        for i in range(self.lib.demon_num_layers(self.h)):
            rc = self.lib.demon_plan_get(self.h, int(n), i, name, 128, ctypes.byref(kind), ctypes.byref(tile), ctypes.byref(ks))
            if rc == 0:
                out[name.value.decode()] = [kind.value, tile.value, ks.value]
        return out

    def set_plan(self, n, plan):
        for layer, (kind, tile, ks) in plan.items():
            self._check(self.lib.demon_plan_set(self.h, int(n), layer.encode(), int(kind), int(tile), int(ks)))

    def set_cu_mask(self, mask_words):
        """demon_set_cu_mask: the context's streams re-created on a compute-unit mask (list of 32-bit words, demon_amd.lanes.cu_masks;
        None / []: every CU again).  Nothing may be in flight; cached graphs are dropped."""
        m = list(mask_words or [])
        arr = (ctypes.c_uint32 * max(1, len(m)))(*m)
        self._check(self.lib.demon_set_cu_mask(self.h, arr, len(m)))

    def load_tuned_plan(self, n, directory=None, nearest=True, lanes=1, partitions=1):
        """installs demon_amd/tuned/plan_<H>x<W>_n<n>.json; without a plan for exactly this batch the plan of the NEAREST tuned
        batch size of the same shape (by ratio) is used -- its kernel families and tiles transfer, and kernels clamp a split-K
        that does not fit -- instead of the untuned heuristics.  Returns the batch size of the plan installed (== n for an
        exact hit), or 0 when none exists for this shape.  lanes > 1 (the context is one lane of a group, demon_amd/lanes.py):
        a plan tuned in throughput mode (tools/tune.py --lanes: plan_..._n<N>_l<L>.json) is preferred when one exists for the
        batch size.  partitions > 1 (the lanes of the group run on disjoint CU masks, LaneGroup(partitions=...)): a plan tuned on a
        1 / partitions share of the compute units (tools/tune.py --cu-partitions: plan_..._n<N>_p<P>.json) is preferred over both."""
        import glob
        import json
        import os
        import re
        directory = directory or os.environ.get("DEMON_PLAN_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned")
        tag = "" if self.version == 1 else "v2_"
        stem = "plan_%s%dx%d_n" % (tag, self.H, self.W)
        have = {}
        part = {}
        if partitions > 1:
            for path in glob.glob(os.path.join(directory, stem + "*_p%d.json" % int(partitions))):
                m = re.match(re.escape(stem) + r"(\d+)_p\d+\.json$", os.path.basename(path))
                if m:
                    part[int(m.group(1))] = path
        for path in glob.glob(os.path.join(directory, stem + "*.json")):
            m = re.match(re.escape(stem) + r"(\d+)\.json$", os.path.basename(path))
            if m:
                have[int(m.group(1))] = path
        if lanes > 1:   # throughput-mode plans replace the latency plan of the same batch size: per batch size the plan tuned for
            tuned = {}  # exactly `lanes` passes in flight, else the one tuned for the nearest lane count
            for path in sorted(glob.glob(os.path.join(directory, stem + "*_l*.json"))):
                m = re.match(re.escape(stem) + r"(\d+)_l(\d+)\.json$", os.path.basename(path))
                if m:
                    tuned.setdefault(int(m.group(1)), {})[int(m.group(2))] = path
            for b, by_l in tuned.items():
                have[b] = by_l[min(by_l, key=lambda l: (abs(l - int(lanes)), l))]
        have.update(part)
        if not have or (int(n) not in have and not nearest):
            return 0
        src = int(n) if int(n) in have else min(have, key=lambda b: abs(np.log(b / float(n))))
        with open(have[src]) as f:
            plan = json.load(f)["plan"]
        self.clear_plan(n)      # a layer only the previous plan mentioned (e.g. a kind-14 marker) must not survive the swap
        self.set_plan(n, plan)
        self.plan_file = os.path.basename(have[src])
        return src

    def set_option(self, key, value):
        self._check(self.lib.demon_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = ctypes.c_int()
        self._check(self.lib.demon_get_option(self.h, key.encode(), ctypes.byref(v)))
        return int(v.value)

    def check_guards(self):
        """demon_debug_check_guards (contexts created under DEMON_POISON_GUARD=1): number of canary zones a kernel wrote into"""
        v = ctypes.c_int()
        self._check(self.lib.demon_debug_check_guards(self.h, ctypes.byref(v)))
        return int(v.value), self.lib.demon_last_error(self.h).decode()

    def clear_plan(self, n):
        """demon_plan_clear: forgets every plan entry of batch size n (before another plan for the same n is installed)"""
        self._check(self.lib.demon_plan_clear(self.h, int(n)))

    # ---- networks (NCHW numpy in, dict of NCHW numpy out) -------------------------------------------------
    def _alloc_outputs(self, n):
        shapes = {
            "predict_flow5": (n, 2, self.h5, self.w5), "predict_conf5": (n, 2, self.h5, self.w5),
            "predict_flow2": (n, 2, self.h2, self.w2), "predict_conf2": (n, 2, self.h2, self.w2),
            "predict_depth2": (n, 1, self.h2, self.w2), "predict_normal2": (n, 3, self.h2, self.w2),
            "predict_rotation": (n, 3), "predict_translation": (n, 3), "predict_scale": (n, 1),
        }
        arrays = {k: np.empty(s, np.float32) for k, s in shapes.items()}
        o = DemonOutputs(**{k: _fp(v) for k, v in arrays.items()})
        return arrays, o

    def bootstrap(self, image_pair, image2_2):
        n = int(np.shape(image_pair)[0])
        image_pair = _f32(image_pair, (n, 6, self.H, self.W), "image_pair")
        image2_2 = _f32(image2_2, (n, 3, self.h2, self.w2), "image2_2")
        arrays, o = self._alloc_outputs(n)
        self._check(self.lib.demon_bootstrap(self.h, n, _fp(image_pair), _fp(image2_2), ctypes.byref(o)))
        return arrays

    def iterative(self, image_pair, image2_2, depth2, normal2, rotation, translation):
        n = int(np.shape(image_pair)[0])
        image_pair = _f32(image_pair, (n, 6, self.H, self.W), "image_pair")
        image2_2 = _f32(image2_2, (n, 3, self.h2, self.w2), "image2_2")
        depth2 = _f32(depth2, (n, 1, self.h2, self.w2), "depth2")
        normal2 = _f32(normal2, (n, 3, self.h2, self.w2), "normal2")
        rotation = _f32(rotation, (n, 3), "rotation")
        translation = _f32(translation, (n, 3), "translation")
        arrays, o = self._alloc_outputs(n)
        self._check(self.lib.demon_iterative(self.h, n, _fp(image_pair), _fp(image2_2), _fp(depth2), _fp(normal2),
                                             _fp(rotation), _fp(translation), ctypes.byref(o)))
        return arrays

    def refine(self, image1, depth2):
        n = int(np.shape(image1)[0])
        image1 = _f32(image1, (n, 3, self.H, self.W), "image1")
        depth2 = _f32(depth2, (n, 1, self.h2, self.w2), "depth2")
        d0 = np.empty((n, 1, self.H, self.W), np.float32)
        self._check(self.lib.demon_refine(self.h, n, _fp(image1), _fp(depth2), _fp(d0)))
        out = {"predict_depth0": d0}
        self._add_normal0(out, n)
        return out

    def _add_normal0(self, out, n):
        if self.version == 2:   # v2/networks.py:223-226
            n0 = np.empty((n, 3, self.H, self.W), np.float32)
            self._check(self.lib.demon_download_normal0(self.h, n, _fp(n0)))
            out["predict_normal0"] = n0

    def full(self, image_pair, image2_2, iterations=3):
        n = int(np.shape(image_pair)[0])
        image_pair = _f32(image_pair, (n, 6, self.H, self.W), "image_pair")
        image2_2 = _f32(image2_2, (n, 3, self.h2, self.w2), "image2_2")
        arrays, o = self._alloc_outputs(n)
        d0 = np.empty((n, 1, self.H, self.W), np.float32)
        self._check(self.lib.demon_full(self.h, n, _fp(image_pair), _fp(image2_2), iterations, ctypes.byref(o), _fp(d0)))
        arrays["predict_depth0"] = d0
        self._add_normal0(arrays, n)
        return arrays

    # ---- device-resident path -----------------------------------------------------------------------------
    def upload_inputs(self, image_pair, image2_2):
        n = int(np.shape(image_pair)[0])
        image_pair = _f32(image_pair, (n, 6, self.H, self.W), "image_pair")
        image2_2 = _f32(image2_2, (n, 3, self.h2, self.w2), "image2_2")
        self._check(self.lib.demon_upload_inputs(self.h, n, _fp(image_pair), _fp(image2_2)))
        return n

    def run_full(self, n, iterations=3):
        self._check(self.lib.demon_run_full(self.h, n, iterations))

    def run_bootstrap(self, n):
        self._check(self.lib.demon_run_bootstrap(self.h, n))

    def synchronize(self):
        self._check(self.lib.demon_synchronize(self.h))

    def release_streams(self):
        """gives the context's HIP streams back (nothing may be in flight); acquire_streams() before the next use"""
        self._check(self.lib.demon_release_streams(self.h))

    def acquire_streams(self):
        self._check(self.lib.demon_acquire_streams(self.h))

    def download_outputs(self, n, with_depth0=True):
        arrays, o = self._alloc_outputs(n)
        d0 = np.empty((n, 1, self.H, self.W), np.float32) if with_depth0 else None
        self._check(self.lib.demon_download_outputs(self.h, n, ctypes.byref(o), _fp(d0) if with_depth0 else None))
        if with_depth0:
            arrays["predict_depth0"] = d0
            self._add_normal0(arrays, n)
        return arrays

    def time_full(self, n, iterations, steps):
        ms = ctypes.c_float()
        self._check(self.lib.demon_time_full(self.h, n, iterations, steps, ctypes.byref(ms)))
        return float(ms.value)

    def profile_full(self, n, iterations=3, repeats=3):
        cap = 1024
        rec = (LaunchRecord * cap)()
        cnt = ctypes.c_int()
        self._check(self.lib.demon_profile_full(self.h, n, iterations, repeats, rec, cap, ctypes.byref(cnt)))
        return [{"name": r.name.decode(), "kernel": r.kernel.decode(), "flops": r.flops, "bytes": r.bytes, "ms": r.ms,
                 "reduce_ms": r.reduce_ms} for r in rec[:min(cnt.value, cap)]]

    # ---- lmbspecialops-level ops --------------------------------------------------------------------------
    def depth_to_flow(self, depth, intrinsics, rotation, translation, inverse_depth=False, normalize_flow=False,
                      gate=False):
        depth = _f32(depth)
        n, c, h, w = depth.shape
        intrinsics = _f32(np.broadcast_to(np.asarray(intrinsics, np.float32), (n, 4)))
        rotation, translation = _f32(rotation, (n, 3)), _f32(translation, (n, 3))
        out = np.empty((n, 2, h, w), np.float32)
        self._check(self.lib.demon_op_depth_to_flow(self.h, _fp(out), _fp(depth), _fp(intrinsics), _fp(rotation),
                                                    _fp(translation), n, h, w, int(inverse_depth), int(normalize_flow),
                                                    int(gate)))
        return out

    def flow_to_depth(self, flow, intrinsics, rotation, translation, inverse_depth=False, normalized_flow=False,
                      method=0):
        flow = _f32(flow)
        n, c, h, w = flow.shape
        intrinsics = _f32(np.broadcast_to(np.asarray(intrinsics, np.float32), (n, 4)))
        rotation, translation = _f32(rotation, (n, 3)), _f32(translation, (n, 3))
        out = np.empty((n, 1, h, w), np.float32)
        self._check(self.lib.demon_op_flow_to_depth(self.h, _fp(out), _fp(flow), _fp(intrinsics), _fp(rotation),
                                                    _fp(translation), n, h, w, int(inverse_depth), int(normalized_flow),
                                                    int(method)))
        return out

    def warp2d(self, inp, displacements, normalized=False, border_mode="clamp", border_value=0.0):
        inp = _f32(inp)
        n, c, h, w = inp.shape
        displacements = _f32(displacements, (n, 2, h, w), "displacements")
        out = np.empty_like(inp)
        self._check(self.lib.demon_op_warp2d(self.h, _fp(out), _fp(inp), _fp(displacements), n, c, h, w, int(normalized),
                                             1 if border_mode == "value" else 0, float(border_value)))
        return out

    def leaky_relu(self, x, leak=0.1):
        x = _f32(x)
        out = np.empty_like(x)
        self._check(self.lib.demon_op_leaky_relu(self.h, _fp(out), _fp(x), x.size, float(leak)))
        return out

    def replace_nonfinite(self, x, value=0.0):
        x = _f32(x)
        out = np.empty_like(x)
        self._check(self.lib.demon_op_replace_nonfinite(self.h, _fp(out), _fp(x), x.size, float(value)))
        return out

    def scale_invariant_gradient(self, x, deltas=(1,), weights=(1.0,), epsilon=0.001):
        """lmbspecialops contract: [N,C,H,W] -> [N*C, 2, H, W] (channels fold into the batch; channel 0 = x, 1 = y) and the
        deltas of one call are SUMMED with their weights -- which is why the reference calls the op once per delta and
        concatenates on axis 1 itself (v2/losses.py:76-79) before slicing (x, y) pairs (:99-102)"""
        x = _f32(x)
        n, c, h, w = x.shape
        d = np.ascontiguousarray(deltas, np.int32)
        wt = _f32(weights, (len(d),), "weights")
        out = np.empty((n * c, 2, h, w), np.float32)
        self._check(self.lib.demon_op_scale_invariant_gradient(
            self.h, _fp(out), _fp(x), n * c, h, w, d.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fp(wt), len(d),
            float(epsilon)))
        return out

    def depth_to_normals(self, depth, intrinsics, inverse_depth=False):
        """v2/losses.py:336-337: [N,1,H,W] depth -> [N,3,H,W] camera-frame normals (NaN at the border / invalid depth).
        UNVERIFIED against lmbspecialops (its source is not in the reference tree): pixel centres at +0.5, one-sided differences
        with the smaller depth change per axis and a normal that points towards the camera are decisions of this library, held
        only by analytic plane tests (tests/test_pins.py) -- normal0 / normal2 ground truth may differ from the reference's."""
        depth = _f32(depth)
        n, c, h, w = depth.shape
        if c != 1:
            raise DemonError("depth_to_normals: depth must have one channel")
        intrinsics = _f32(np.broadcast_to(np.asarray(intrinsics, np.float32), (n, 4)))
        out = np.empty((n, 3, h, w), np.float32)
        self._check(self.lib.demon_op_depth_to_normals(self.h, _fp(out), _fp(depth), _fp(intrinsics), n, h, w, int(inverse_depth)))
        return out

    def median3x3_downsample(self, x):
        x = _f32(x)
        n, c, h, w = x.shape
        out = np.empty((n, c, (h + 1) // 2, (w + 1) // 2), np.float32)
        self._check(self.lib.demon_op_median3x3_downsample(self.h, _fp(out), _fp(x), n * c, h, w))
        return out

    def pointwise_l2_loss(self, inp, gt, epsilon):
        """v2/losses.py:33-54 (NCHW): mean over pixels of sqrt(sum_c replace_nonfinite(inp - gt)^2 + epsilon)"""
        inp = _f32(inp)
        gt = _f32(gt, inp.shape, "gt")
        n, c, h, w = inp.shape
        out = ctypes.c_float()
        self._check(self.lib.demon_op_pointwise_l2_loss(self.h, ctypes.byref(out), _fp(inp), _fp(gt), n, c, h, w, float(epsilon)))
        return float(out.value)

    # ---- single layers (TF weight layouts) ----------------------------------------------------------------
    def conv2d(self, x, w_hwio, bias, stride=(1, 1), lrelu=False, padding="caffe"):
        """padding 'caffe': k//2 zeros on both sides then VALID (helpers.py:70-94); 'same': tf.layers.conv2d(padding='same')
        (v2/helpers.py:24-35)"""
        x, w_hwio, bias = _f32(x), _f32(w_hwio), _f32(bias)
        n, cin, h, w = x.shape
        kh, kw, ci, cout = w_hwio.shape
        if ci != cin or bias.shape != (cout,):
            raise DemonError("conv2d: weight / bias shape mismatch")
        sh, sw = stride
        if padding == "same":
            ho, wo, ph, pw = -(-h // sh), -(-w // sw), -1, -1
        elif padding == "caffe":
            ho, wo, ph, pw = (h + 2 * (kh // 2) - kh) // sh + 1, (w + 2 * (kw // 2) - kw) // sw + 1, kh // 2, kw // 2
        else:
            raise DemonError("padding must be 'caffe' or 'same'")
        out = np.empty((n, cout, ho, wo), np.float32)
        self._check(self.lib.demon_op_conv2d(self.h, _fp(out), _fp(x), _fp(w_hwio), _fp(bias), n, cin, h, w, cout, kh, kw,
                                             sh, sw, ph, pw, int(lrelu)))
        return out

    def deconv4x4s2(self, x, w_hwoi, bias, lrelu=False):
        x, w_hwoi, bias = _f32(x), _f32(w_hwoi), _f32(bias)
        n, cin, h, w = x.shape
        if w_hwoi.shape[:2] != (4, 4) or w_hwoi.shape[3] != cin:
            raise DemonError("deconv4x4s2: weight must be [4,4,Cout,Cin]")
        cout = w_hwoi.shape[2]
        out = np.empty((n, cout, 2 * h, 2 * w), np.float32)
        self._check(self.lib.demon_op_deconv4x4s2(self.h, _fp(out), _fp(x), _fp(w_hwoi), _fp(bias), n, cin, h, w, cout,
                                                  int(lrelu)))
        return out

    def dense(self, x, w_io, bias, lrelu=False):
        x, w_io, bias = _f32(x), _f32(w_io), _f32(bias)
        n, cin = x.shape
        cout = w_io.shape[1]
        out = np.empty((n, cout), np.float32)
        self._check(self.lib.demon_op_dense(self.h, _fp(out), _fp(x), _fp(w_io), _fp(bias), n, cin, cout, int(lrelu)))
        return out

    # ---- tuning / diagnostics -----------------------------------------------------------------------------
    def last_kernel(self):
        """tag of the contraction kernel this thread launched last (diagnostic; e.g. 'wino_deconv<16x64>+splitk')"""
        buf = ctypes.create_string_buffer(64)
        self.lib.demon_last_kernel(buf, 64)
        return buf.value.decode()

    def bench_layer(self, kind, n, cin, h, w, cout, kh=1, kw=1, sh=1, sw=1, tile=-1, ksplit=0, iters=20):
        """kind: 'conv' | 'deconv' | 'dense'.  Returns (avg_ms, TFLOP/s)."""
        k = {"conv": 0, "deconv": 1, "dense": 2}[kind]
        ms = ctypes.c_float()
        fl = ctypes.c_double()
        self._check(self.lib.demon_bench_layer(self.h, k, n, cin, h, w, cout, kh, kw, sh, sw, tile, ksplit, iters,
                                               ctypes.byref(ms), ctypes.byref(fl)))
        return float(ms.value), fl.value / (ms.value * 1e-3) / 1e12
