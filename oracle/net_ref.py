"""CPU restatement of the DeMoN sub-networks -- TEST INFRASTRUCTURE ONLY ("TF-CPU-equivalent").

Topology follows python/depthmotionnet/blocks_original.py (flow block :121-235, depth+motion block
:299-448, refinement block :452-513) and the layer helpers python/depthmotionnet/helpers.py:70-153;
feeds / fetches follow python/depthmotionnet/networks_original.py:38-57, :108-152, :219-234.
Convolution arithmetic (TensorFlow 1.4 in the reference, not available here) is done by PyTorch-CPU
F.conv2d / F.conv_transpose2d / F.linear on float32 with the TF->torch weight permutations of
SURVEY.md appendix B/D; those are cross-checked at small sizes against the naive double-accumulating
C loops in oracle/demon_oracle.c (tests/test_oracle.py).  lmbspecialops calls go to the C restatement
(oracle/ops_ref.py).

weights: dict  TF variable name -> numpy array in TF layout
   conv   '<scope>/<name>/kernel' [kh,kw,Cin,Cout], deconv [4,4,Cout,Cin], dense [in,out], '.../bias' [Cout]
All activations are NCHW ('channels_first'); the API edge converts for 'channels_last'.

The v2 model (python/depthmotionnet/v2/blocks.py, v2/helpers.py, v2/networks.py) is restated below the
original one: the same Net helper with TensorFlow's padding='same' rule written out in full
(out = ceil(n / s); pad_total = max((out - 1) * s + k - n, 0); pad_before = pad_total // 2).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops_ref

INTRINSICS = np.array([0.89115971, 1.18821287, 0.5, 0.5], np.float32)  # networks_original.py:108


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def lrelu(x):
    # helpers.py:60-63
    return torch.where(x >= 0, x, 0.1 * x)


class Net:
    """Holds the torch-layout weights of one variable scope ('netFlow1', ...)."""

    def __init__(self, weights, scope, same=False):
        self.w = weights
        self.scope = scope
        self.same = same     # False: helpers.py caffe padding; True: v2/helpers.py padding='same'
        self._cache = {}

    def _pad(self, x, kh, kw, sh, sw):
        if not self.same:
            return F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2))
        pads = []
        for n, k, s in ((x.shape[3], kw, sw), (x.shape[2], kh, sh)):   # F.pad order: W first
            out = -(-n // s)
            total = max((out - 1) * s + k - n, 0)
            pads += [total // 2, total - total // 2]
        return F.pad(x, tuple(pads))

    def _conv_w(self, name):
        key = ("c", name)
        if key not in self._cache:
            k = self.w["%s/%s/kernel" % (self.scope, name)]  # HWIO
            self._cache[key] = (_t(k.transpose(3, 2, 0, 1)), _t(self.w["%s/%s/bias" % (self.scope, name)]))
        return self._cache[key]

    def _deconv_w(self, name):
        key = ("d", name)
        if key not in self._cache:
            k = self.w["%s/%s/kernel" % (self.scope, name)]  # [kh,kw,Cout,Cin]
            self._cache[key] = (_t(k.transpose(3, 2, 0, 1)), _t(self.w["%s/%s/bias" % (self.scope, name)]))
        return self._cache[key]

    def _dense_w(self, name):
        key = ("f", name)
        if key not in self._cache:
            k = self.w["%s/%s/kernel" % (self.scope, name)]  # [in,out]
            self._cache[key] = (_t(k.T), _t(self.w["%s/%s/bias" % (self.scope, name)]))
        return self._cache[key]

    # helpers.py:70-94 (+ :97-102 with activation)
    def conv(self, x, name, stride=1, act=True):
        w, b = self._conv_w(name)
        kh, kw = w.shape[2], w.shape[3]
        y = F.conv2d(self._pad(x, kh, kw, stride, stride), w, b, stride=stride)
        return lrelu(y) if act else y

    # helpers.py:105-153: k x 1 (stride (s,1)) then 1 x k (stride (1,s)), both leaky relu
    def conv2(self, x, name, k, stride):
        wy, by = self._conv_w(name + "y")
        wx, bx = self._conv_w(name + "x")
        t = lrelu(F.conv2d(self._pad(x, k, 1, stride, 1), wy, by, stride=(stride, 1)))
        return lrelu(F.conv2d(self._pad(t, 1, k, 1, stride), wx, bx, stride=(1, stride)))

    # blocks_original.py:97-110: deconv k4 s2 VALID, activation, crop 1 -> padding=1 in torch
    # blocks_original.py:64-75 ('same', linear) has the same geometry
    def deconv(self, x, name, act=True):
        w, b = self._deconv_w(name)
        y = F.conv_transpose2d(x, w, b, stride=2, padding=1)
        return lrelu(y) if act else y

    def dense(self, x, name, act=True):
        w, b = self._dense_w(name)
        y = F.linear(x, w, b)
        return lrelu(y) if act else y


def flow_block(net, image_pair, image2_2=None, prev=None):
    """blocks_original.py:121-235.  prev = dict(depth2, normal2, rotation, translation) or None."""
    conv1 = net.conv2(image_pair, "conv1", 9, 2)
    if prev is None:
        conv2 = net.conv2(conv1, "conv2", 7, 2)          # 64 outputs in netFlow1 (:144)
        conv2_1 = net.conv2(conv2, "conv2_1", 3, 1)
    else:
        conv2 = net.conv2(conv1, "conv2", 7, 2)          # 32 outputs (:147)
        N = image_pair.shape[0]
        flow = ops_ref.depth_to_flow(prev["depth2"].numpy(), INTRINSICS, prev["rotation"].numpy(),
                                     prev["translation"].numpy(), inverse_depth=True, normalize_flow=True,
                                     gate=True)          # :155-168
        warped = ops_ref.warp2d(image2_2.numpy(), flow, normalized=True, border_mode="value")  # :171-176
        extra = torch.cat((_t(warped), _t(flow), prev["depth2"], prev["normal2"]), 1)          # :180-183
        conv_extra = net.conv2(extra, "conv2_extra_inputs", 3, 1)
        conv2_1 = net.conv2(torch.cat((conv2, conv_extra), 1), "conv2_1", 3, 1)                # :186-187
    conv3 = net.conv2(conv2_1, "conv3", 5, 2)
    conv3_1 = net.conv2(conv3, "conv3_1", 3, 1)
    conv4 = net.conv2(conv3_1, "conv4", 5, 2)
    conv4_1 = net.conv2(conv4, "conv4_1", 3, 1)
    conv5 = net.conv2(conv4_1, "conv5", 5, 2)
    conv5_1 = net.conv2(conv5, "conv5_1", 3, 1)
    flowconf5 = net.conv(net.conv(conv5_1, "predict_flow5/conv1"), "predict_flow5/conv2", act=False)
    up5to4 = net.deconv(flowconf5, "upsample_flow5to4/upconv", act=False)
    concat4 = torch.cat((net.deconv(conv5_1, "refine4/upconv"), conv4_1, up5to4), 1)           # :111
    concat3 = torch.cat((net.deconv(concat4, "refine3/upconv"), conv3_1), 1)
    concat2 = torch.cat((net.deconv(concat3, "refine2/upconv"), conv2_1), 1)
    flowconf2 = net.conv(net.conv(concat2, "predict_flow2/conv1"), "predict_flow2/conv2", act=False)
    return {"predict_flowconf5": flowconf5, "predict_flowconf2": flowconf2}


def depthmotion_block(net, image_pair, image2_2, flowconf2, prev_rt=None, flow_to_depth_method=0):
    """blocks_original.py:299-448.  prev_rt = (rotation, translation) of the previous iteration or None."""
    conv1 = net.conv2(image_pair, "conv1", 9, 2)
    conv2 = net.conv2(conv1, "conv2", 7, 2)
    flow2 = flowconf2[:, 0:2].contiguous()
    warped = ops_ref.warp2d(image2_2.numpy(), flow2.numpy(), normalized=True, border_mode="value")  # :336
    extra = [_t(warped), flowconf2]
    if prev_rt is not None:
        d = ops_ref.flow_to_depth(flow2.numpy(), INTRINSICS, prev_rt[0].numpy(), prev_rt[1].numpy(),
                                  inverse_depth=True, normalized_flow=True, method=flow_to_depth_method)  # :344
        extra.append(_t(d))
    conv_extra = net.conv2(torch.cat(extra, 1), "conv2_extra_inputs", 3, 1)
    conv2_1 = net.conv2(torch.cat((conv2, conv_extra), 1), "conv2_1", 3, 1)
    conv3 = net.conv2(conv2_1, "conv3", 5, 2)
    conv3_1 = net.conv2(conv3, "conv3_1", 3, 1)
    conv4 = net.conv2(conv3_1, "conv4", 5, 2)
    conv4_1 = net.conv2(conv4, "conv4_1", 3, 1)
    conv5 = net.conv2(conv4_1, "conv5", 3, 2)            # k=3 in the DM nets (:375)
    conv5_1 = net.conv2(conv5, "conv5_1", 3, 1)
    motion_conv1 = net.conv(conv5_1, "motion_conv1")
    fc = motion_conv1.reshape(motion_conv1.shape[0], -1)  # flatten in C,H,W order (:388-392)
    fc = net.dense(net.dense(fc, "motion_fc1"), "motion_fc2")
    motion = net.dense(fc, "motion_fc3", act=False)
    rotation, translation, scale = motion[:, 0:3], motion[:, 3:6], motion[:, 6:7]   # :412
    concat4 = torch.cat((net.deconv(conv5_1, "refine4/upconv"), conv4_1), 1)
    concat3 = torch.cat((net.deconv(concat4, "refine3/upconv"), conv3_1), 1)
    concat2 = torch.cat((net.deconv(concat3, "refine2/upconv"), conv2_1), 1)
    dn = net.conv(net.conv(concat2, "predict_depthnormal2/conv1"), "predict_depthnormal2/conv2", act=False)
    depth = scale.reshape(-1, 1, 1, 1) * dn[:, 0:1]     # :278-283: only the depth channel is scaled
    return {"predict_depth2": depth.contiguous(), "predict_normal2": dn[:, 1:4].contiguous(),
            "predict_rotation": rotation.contiguous(), "predict_translation": translation.contiguous(),
            "predict_scale": scale.contiguous()}


def refine_block(net, image1, depth2):
    """blocks_original.py:452-513."""
    H, W = image1.shape[2], image1.shape[3]
    up = _t(ops_ref.resize_nearest(depth2.numpy(), H, W))          # :475
    x = torch.cat((image1, up), 1)                                   # :482
    conv0 = net.conv(x, "conv0")
    conv1 = net.conv(conv0, "conv1", stride=2)
    conv1_1 = net.conv(conv1, "conv1_1")
    conv2 = net.conv(conv1_1, "conv2", stride=2)
    conv2_1 = net.conv(conv2, "conv2_1")
    concat1 = torch.cat((net.deconv(conv2_1, "refine1/upconv"), conv1_1), 1)
    concat0 = torch.cat((net.deconv(concat1, "refine0/upconv"), conv0), 1)
    d0 = net.conv(net.conv(concat0, "predict_depth0/conv1"), "predict_depth0/conv2", act=False)
    return {"predict_depth0": d0}


class DemonRef:
    """The five sub-nets with the stage order of examples/example.py:87-99."""

    def __init__(self, weights, flow_to_depth_method=0):
        self.nets = {s: Net(weights, s) for s in ("netFlow1", "netDM1", "netFlow2", "netDM2", "netRefine")}
        self.method = flow_to_depth_method

    @torch.no_grad()
    def bootstrap(self, image_pair, image2_2):
        image_pair, image2_2 = _t(image_pair), _t(image2_2)
        f = flow_block(self.nets["netFlow1"], image_pair)
        dm = depthmotion_block(self.nets["netDM1"], image_pair, image2_2, f["predict_flowconf2"])
        return self._pack(f, dm)

    @torch.no_grad()
    def iterative(self, image_pair, image2_2, depth2, normal2, rotation, translation):
        image_pair, image2_2 = _t(image_pair), _t(image2_2)
        prev = {"depth2": _t(depth2), "normal2": _t(normal2), "rotation": _t(rotation),
                "translation": _t(translation)}
        f = flow_block(self.nets["netFlow2"], image_pair, image2_2, prev)
        dm = depthmotion_block(self.nets["netDM2"], image_pair, image2_2, f["predict_flowconf2"],
                               (prev["rotation"], prev["translation"]), self.method)
        return self._pack(f, dm)

    @torch.no_grad()
    def refine(self, image1, depth2):
        r = refine_block(self.nets["netRefine"], _t(image1), _t(depth2))
        return {"predict_depth0": r["predict_depth0"].numpy()}

    def full(self, image_pair, image2_2, iterations=3):
        r = self.bootstrap(image_pair, image2_2)
        for _ in range(iterations):
            r = self.iterative(image_pair, image2_2, r["predict_depth2"], r["predict_normal2"],
                               r["predict_rotation"], r["predict_translation"])
        out = dict(r)
        out.update(self.refine(np.ascontiguousarray(image_pair[:, 0:3]), r["predict_depth2"]))
        return out

    @staticmethod
    def _pack(f, dm):
        return {
            "predict_flow5": f["predict_flowconf5"][:, 0:2].contiguous().numpy(),
            "predict_conf5": f["predict_flowconf5"][:, 2:4].contiguous().numpy(),
            "predict_flow2": f["predict_flowconf2"][:, 0:2].contiguous().numpy(),
            "predict_conf2": f["predict_flowconf2"][:, 2:4].contiguous().numpy(),
            "predict_depth2": dm["predict_depth2"].numpy(),
            "predict_normal2": dm["predict_normal2"].numpy(),
            "predict_rotation": dm["predict_rotation"].numpy(),
            "predict_translation": dm["predict_translation"].numpy(),
            "predict_scale": dm["predict_scale"].numpy(),
        }


# ======================================================================================================
# v2 model: python/depthmotionnet/v2/blocks.py
# ======================================================================================================
def _dense5(net, conv5_1):
    """v2/blocks.py:197-213, :395-411: first 96 channels, flattened C,H,W, square dense + lrelu, back as 96 channels."""
    n, _, h, w = conv5_1.shape
    d = net.dense(conv5_1[:, 0:96].reshape(n, -1), "dense5")
    return torch.cat((conv5_1, d.reshape(n, 96, h, w)), 1)


def flow_block_v2(net, image_pair, image2_2=None, prev=None):
    """v2/blocks.py:120-253."""
    conv1 = net.conv2(image_pair, "conv1", 9, 2)
    if prev is None:
        conv2 = net.conv2(conv1, "conv2", 7, 2)
        conv2_1 = net.conv2(conv2, "conv2_1", 3, 1)
    else:
        conv2 = net.conv2(conv1, "conv2", 7, 2)
        flow = ops_ref.depth_to_flow(prev["depth2"].numpy(), INTRINSICS, prev["rotation"].numpy(),
                                     prev["translation"].numpy(), inverse_depth=True, normalize_flow=True,
                                     gate=True)          # :155-168
        warped = ops_ref.warp2d(image2_2.numpy(), flow, normalized=True, border_mode="value")  # :171-176
        extra = torch.cat((_t(warped), _t(flow), prev["depth2"], prev["normal2"]), 1)          # :180-183
        conv_extra = net.conv2(extra, "conv2_extra_inputs", 3, 1)
        conv2_1 = net.conv2(torch.cat((conv2, conv_extra), 1), "conv2_1", 3, 1)                # :186-187
    conv3_1 = net.conv2(net.conv2(conv2_1, "conv3", 5, 2), "conv3_1", 3, 1)
    conv4_1 = net.conv2(net.conv2(conv3_1, "conv4", 5, 2), "conv4_1", 3, 1)
    conv5_1 = net.conv2(net.conv2(conv4_1, "conv5", 5, 2), "conv5_1", 3, 1)
    feat5 = _dense5(net, conv5_1)
    flowconf5 = net.conv(net.conv(feat5, "predict_flow5/conv1"), "predict_flow5/conv2", act=False)
    up5to4 = net.deconv(flowconf5, "upsample_flow5to4/upconv", act=False)
    concat4 = torch.cat((net.deconv(feat5, "refine4/upconv"), conv4_1, up5to4), 1)             # :110-116 order
    concat3 = torch.cat((net.deconv(concat4, "refine3/upconv"), conv3_1), 1)
    concat2 = torch.cat((net.deconv(concat3, "refine2/upconv"), conv2_1), 1)
    flowconf2 = net.conv(net.conv(concat2, "predict_flow2/conv1"), "predict_flow2/conv2", act=False)
    return {"predict_flowconf5": flowconf5, "predict_flowconf2": flowconf2}


def depthmotion_block_v2(net, image_pair, image2_2, flowconf2, prev_rt=None):
    """v2/blocks.py:317-494."""
    conv1 = net.conv2(image_pair, "conv1", 9, 2)
    conv2 = net.conv2(conv1, "conv2", 7, 2)
    flow2 = flowconf2[:, 0:2].contiguous()
    warped = ops_ref.warp2d(image2_2.numpy(), flow2.numpy(), normalized=True, border_mode="value")  # :351
    extra = [_t(warped), flowconf2]
    if prev_rt is not None:
        d = ops_ref.flow_to_depth2(flow2.numpy(), INTRINSICS, prev_rt[0].numpy(), prev_rt[1].numpy(),
                                   inverse_depth=True, normalized_flow=True)                    # :362-369
        # :379 clip_by_value(0, 50) = max(min(x, 50), 0); NaN -> 50 as with the fminf / fmaxf of TF's GPU kernels
        extra.append(_t(np.fmax(np.fmin(d, np.float32(50.0)), np.float32(0.0))))
    conv_extra = net.conv2(torch.cat(extra, 1), "conv2_extra_inputs", 3, 1)
    conv2_1 = net.conv2(torch.cat((conv2, conv_extra), 1), "conv2_1", 3, 1)
    conv3_1 = net.conv2(net.conv2(conv2_1, "conv3", 5, 2), "conv3_1", 3, 1)
    conv4_1 = net.conv2(net.conv2(conv3_1, "conv4", 5, 2), "conv4_1", 3, 1)
    conv5_1 = net.conv2(net.conv2(conv4_1, "conv5", 3, 2), "conv5_1", 3, 1)                     # k=3 (:392)
    feat5 = _dense5(net, conv5_1)
    m = net.conv2(net.conv2(conv2_1, "motion_conv3", 5, 2), "motion_conv4", 5, 2)               # :414-415
    m5a = net.conv2(m, "motion_conv5a", 3, 2)
    m5b = net.conv(feat5, "motion_conv5b")
    fc = torch.cat((m5a, m5b), 1)
    fc = fc.reshape(fc.shape[0], -1)                                                           # :431
    fc = net.dense(net.dense(fc, "motion_fc1"), "motion_fc2")
    motion = net.dense(fc, "motion_fc3", act=False)
    rotation, translation, scale = motion[:, 0:3], motion[:, 3:6], motion[:, 6:7]               # :457
    concat4 = torch.cat((net.deconv(conv5_1, "refine4/upconv"), conv4_1), 1)                    # conv5_1, not feat5 (:462)
    concat3 = torch.cat((net.deconv(concat4, "refine3/upconv"), conv3_1), 1)
    concat2 = torch.cat((net.deconv(concat3, "refine2/upconv"), conv2_1), 1)
    dn = net.conv(net.conv(concat2, "predict_depthnormal2/conv1"), "predict_depthnormal2/conv2", act=False)
    depth = scale.reshape(-1, 1, 1, 1) * dn[:, 0:1]                                             # :294-300
    return {"predict_depth2": depth.contiguous(), "predict_normal2": dn[:, 1:4].contiguous(),
            "predict_rotation": rotation.contiguous(), "predict_translation": translation.contiguous(),
            "predict_scale": scale.contiguous()}


def refine_block_v2(net, image1, depth2):
    """v2/blocks.py:499-562."""
    H, W = image1.shape[2], image1.shape[3]
    up = _t(ops_ref.resize_nearest(depth2.numpy(), H, W))
    x = torch.cat((image1, up), 1)
    conv0 = net.conv(x, "conv0")
    conv1_1 = net.conv(net.conv(conv0, "conv1", stride=2), "conv1_1")
    conv2_1 = net.conv(net.conv(conv1_1, "conv2", stride=2), "conv2_1")
    concat1 = torch.cat((net.deconv(conv2_1, "refine1/upconv"), conv1_1), 1)
    concat0 = torch.cat((net.deconv(concat1, "refine0/upconv"), conv0), 1)
    dn = net.conv(net.conv(concat0, "predict_depth0/conv1"), "predict_depth0/conv2", act=False)
    return {"predict_depth0": dn[:, 0:1].contiguous(), "predict_normal0": dn[:, 1:4].contiguous()}


class DemonRefV2(DemonRef):
    """The v2 sub-nets with the stage order of examples/example_v2.py:93-105."""

    def __init__(self, weights):
        self.nets = {s: Net(weights, s, same=True) for s in ("netFlow1", "netDM1", "netFlow2", "netDM2", "netRefine")}

    @torch.no_grad()
    def bootstrap(self, image_pair, image2_2):
        image_pair, image2_2 = _t(image_pair), _t(image2_2)
        f = flow_block_v2(self.nets["netFlow1"], image_pair)
        dm = depthmotion_block_v2(self.nets["netDM1"], image_pair, image2_2, f["predict_flowconf2"])
        return self._pack(f, dm)

    @torch.no_grad()
    def iterative(self, image_pair, image2_2, depth2, normal2, rotation, translation):
        image_pair, image2_2 = _t(image_pair), _t(image2_2)
        prev = {"depth2": _t(depth2), "normal2": _t(normal2), "rotation": _t(rotation),
                "translation": _t(translation)}
        f = flow_block_v2(self.nets["netFlow2"], image_pair, image2_2, prev)
        dm = depthmotion_block_v2(self.nets["netDM2"], image_pair, image2_2, f["predict_flowconf2"],
                                  (prev["rotation"], prev["translation"]))
        return self._pack(f, dm)

    @torch.no_grad()
    def refine(self, image1, depth2):
        r = refine_block_v2(self.nets["netRefine"], _t(image1), _t(depth2))
        return {k: v.numpy() for k, v in r.items()}
