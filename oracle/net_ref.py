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

    def __init__(self, weights, scope):
        self.w = weights
        self.scope = scope
        self._cache = {}

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
        x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2))
        y = F.conv2d(x, w, b, stride=stride)
        return lrelu(y) if act else y

    # helpers.py:105-153: k x 1 (stride (s,1)) then 1 x k (stride (1,s)), both leaky relu
    def conv2(self, x, name, k, stride):
        wy, by = self._conv_w(name + "y")
        wx, bx = self._conv_w(name + "x")
        p = k // 2
        t = lrelu(F.conv2d(F.pad(x, (0, 0, p, p)), wy, by, stride=(stride, 1)))
        return lrelu(F.conv2d(F.pad(t, (p, p, 0, 0)), wx, bx, stride=(1, stride)))

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
