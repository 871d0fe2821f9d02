"""Holds this repo to tests/golden/wiring_original.npz / wiring_v2.npz: the outputs of the REFERENCE'S OWN graph-building code
(python/depthmotionnet/networks_original.py:22-255 over blocks_original.py and helpers.py; v2/networks.py over v2/blocks.py),
executed unmodified in the build container on the TensorFlow-1.4 graph emulator oracle/tf1/ by tests/golden/make_golden_wiring.py.

What these files pin (and SURVEY.md / DESIGN.md could only assert by reading until round 5): the variable table TensorFlow's
scoping rules give the reference code (242 / 274 names and shapes -- the set demon_set_weight accepts and the checkpoint reader
maps), and the wiring of all five sub-nets: layer order, channel counts, concatenation orders, which previous prediction feeds
which op, the crop of the transposed convs, the flatten order in front of motion_fc1, the nearest-neighbour upsampling in front of
the refinement net -- for both data formats (the channels_last graph gave identical numbers, `nhwc_max_abs_diff`).
What they do not pin: the arithmetic inside tf.layers.* / lmbspecialops.* (the emulator restates it; see its header).

  -m "not gpu": the variable table, and oracle/net_ref.py stage by stage (each stage fed the golden outputs of the previous one)
  -m gpu      : the HIP path through the reference's API (BootstrapNet / IterativeNet / RefinementNet of
                python/depthmotionnet, both data formats) and through DemonContext, stage by stage, <= 1e-3 relative L1
"""
import json
import os
import sys

import numpy as np
import pytest

from conftest import rel_l1

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import golden_common as G  # noqa: E402

ORACLE_TOL = 2e-5    # two independent fp32 restatements of the same sums (numpy tensordot by taps vs PyTorch convolutions): observed <= 3e-6
NET_TOL = 1e-3       # BASELINE.json north_star
STAGES = ("bootstrap", "iterative0", "iterative1", "iterative2", "refine")


def _load(version):
    return np.load(os.path.join(ROOT, "tests", "golden", "wiring_original.npz" if version == 1 else "wiring_v2.npz"))


def _inputs(f, case, version):
    if case.startswith("synthetic"):
        seed = int(f["input_seed"]) + (1 if case.endswith("_b") else 0)
        return G.synthetic_pair(int(f["batch"]) if version == 1 else 1, seed)
    return f[case + "/in/image_pair"], f[case + "/in/image2_2"]


def _weights(f, version):
    variables = [(n, tuple(s)) for n, s in json.loads(str(f["variables"]))]
    return G.seeded_weights(variables, int(f["seed"]), consistent_flow=(version == 2))


def _stage(f, case, stage):
    prefix = "%s/out/%s/" % (case, stage)
    return dict((k[len(prefix):], f[k]) for k in f.files if k.startswith(prefix))


def _check(f, version, run, tol):
    """run(stage, pair, img2_2, prev golden stage) -> dict; every golden key of every stage of every case within tol"""
    worst, count = 0.0, 0
    for case in json.loads(str(f["cases"])):
        pair, img2_2 = _inputs(f, case, version)
        prev = None
        for stage in STAGES:
            want = _stage(f, case, stage)
            assert want, (case, stage)
            got = run(stage, pair, img2_2, prev)
            for k, v in want.items():
                assert got[k].shape == v.shape, (case, stage, k, got[k].shape, v.shape)
                err = rel_l1(got[k], v)
                assert np.isfinite(got[k]).all() and err <= tol, "%s %s %s: relative L1 %.3e" % (case, stage, k, err)
                worst, count = max(worst, err), count + 1
            if stage != "refine":
                prev = want
    assert count >= 2 * (4 * 6 + 1)
    return worst


@pytest.mark.parametrize("version", [1, 2])
def test_variable_table_is_the_one_the_reference_code_creates(version):
    from demon_amd import weights as W
    f = _load(version)
    assert "reference graph code" in str(f["backend"]) and float(f["nhwc_max_abs_diff"]) == 0.0
    theirs = [(n, tuple(s)) for n, s in json.loads(str(f["variables"]))]
    ours = W.variable_shapes(version=version)
    assert len(theirs) == len(set(n for n, _ in theirs)) == len(ours) == (242 if version == 1 else 274)
    assert dict(theirs) == dict((k, tuple(v)) for k, v in ours.items())
    assert sum(int(np.prod(s)) for _, s in theirs) == (45753883 if version == 1 else 123219422)
    # scopes: exactly the five sub-nets of networks_original.py:44,50,125,142,227
    assert sorted(set(n.split("/")[0] for n, _ in theirs)) == ["netDM1", "netDM2", "netFlow1", "netFlow2", "netRefine"]


@pytest.mark.parametrize("version", [1, 2])
def test_oracle_equals_the_reference_wiring(version):
    from oracle import net_ref
    f = _load(version)
    w = _weights(f, version)
    net = net_ref.DemonRef(w) if version == 1 else net_ref.DemonRefV2(w)

    def run(stage, pair, img2_2, prev):
        if stage == "bootstrap":
            return net.bootstrap(pair, img2_2)
        if stage == "refine":
            return net.refine(pair[:, 0:3], prev["predict_depth2"])
        return net.iterative(pair, img2_2, prev["predict_depth2"], prev["predict_normal2"], prev["predict_rotation"], prev["predict_translation"])

    assert _check(f, version, run, ORACLE_TOL) > 0.0      # (two different implementations: a bit-identical answer would mean one was compared with itself)


@pytest.mark.gpu
@pytest.mark.parametrize("version", [1, 2])
def test_hip_context_equals_the_reference_wiring(version):
    from demon_amd import DemonContext
    f = _load(version)
    w = _weights(f, version)
    ctx = DemonContext(0, 2, 192, 256, version=version)
    try:
        ctx.set_weights(w)

        def run(stage, pair, img2_2, prev):
            if stage == "bootstrap":
                return ctx.bootstrap(pair, img2_2)
            if stage == "refine":
                return ctx.refine(pair[:, 0:3], prev["predict_depth2"])
            return ctx.iterative(pair, img2_2, prev["predict_depth2"], prev["predict_normal2"], prev["predict_rotation"], prev["predict_translation"])

        _check(f, version, run, NET_TOL)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("data_format", ["channels_first", "channels_last"])
def test_reference_api_equals_the_reference_wiring(data_format, monkeypatch):
    """the drop-in boundary itself: `from depthmotionnet.networks_original import *` of THIS repo (python/depthmotionnet), fed like
    examples/example.py:75-99 feeds the reference's classes, in both data formats"""
    monkeypatch.syspath_prepend(os.path.join(ROOT, "python"))
    from depthmotionnet.networks_original import BootstrapNet, IterativeNet, RefinementNet
    from demon_amd import runtime
    f = _load(1)
    w = _weights(f, 1)

    def to(a):
        return a if data_format == "channels_first" or a.ndim != 4 else np.ascontiguousarray(a.transpose(0, 2, 3, 1))

    def back(d):
        return dict((k, v if data_format == "channels_first" or v.ndim != 4 else v.transpose(0, 3, 1, 2)) for k, v in d.items())

    nets = {}

    def run(stage, pair, img2_2, prev):
        n = pair.shape[0]
        if n not in nets:
            session = type("Session", (), {"demon_weights": None})()      # a session that carries its own variables (python/tf_stub)
            boot, it, ref = BootstrapNet(session, data_format, n), IterativeNet(session, data_format, n), RefinementNet(session, data_format, n)
            runtime.set_session_weights(session, w)                       # = Saver.restore after the constructors (example.py:75-83)
            nets[n] = (session, boot, it, ref)
        _, boot, it, ref = nets[n]
        if stage == "bootstrap":
            return back(boot.eval(to(pair), to(img2_2)))
        if stage == "refine":
            return back(ref.eval(to(pair[:, 0:3]), to(prev["predict_depth2"])))
        return back(it.eval(to(pair), to(img2_2), to(prev["predict_depth2"]), to(prev["predict_normal2"]), prev["predict_rotation"], prev["predict_translation"]))

    try:
        _check(f, 1, run, NET_TOL)
    finally:
        for session, _, _, _ in nets.values():
            for key, ctx in list(runtime._contexts.items()):
                if getattr(ctx, "_owner", None) is session:
                    ctx.close()
                    del runtime._contexts[key]


# ---- the ground-truth preparation / loss helpers of the v2 training code ------------------------------
# tests/golden/wiring_losses.npz = the reference's own v2/losses.py:23-104, :312-374 executed on oracle/tf1 (make_golden_losses.py)
class _OracleSops:
    """demon_amd/sops.py's interface with oracle/ops_ref.py behind it: lets the CPU suite hold demon_amd/losses.py's WIRING to the
    reference's (which op on which level of the pyramid, one scale_invariant_gradient call per delta, NaN replacement, channel
    pairs of the gradient loss) without a GPU"""

    def __getattr__(self, name):
        from oracle import ops_ref
        if name == "_ctx":
            return lambda: self
        if name == "depth_to_flow":
            return lambda depth, intrinsics, rotation, translation, rotation_format="angleaxis3", inverse_depth=False, normalize_flow=False, name=None: \
                ops_ref.depth_to_flow(depth, intrinsics, rotation, translation, inverse_depth, normalize_flow)
        if name == "scale_invariant_gradient":
            return lambda input, deltas=(1,), weights=(1.0,), epsilon=0.001: ops_ref.scale_invariant_gradient(input, deltas, weights, epsilon)
        return getattr(ops_ref, name)


def _losses_inputs():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_losses
    return make_golden_losses.inputs()


def _check_losses(L, tol):
    f = np.load(os.path.join(ROOT, "tests", "golden", "wiring_losses.npz"))
    assert "reference v2/losses.py" in str(f["backend"])
    x = _losses_inputs()
    got = dict(("gt/" + k, v) for k, v in L.prepare_ground_truth_tensors(x["depth"], x["rotation"], x["translation"], x["intrinsics"]).items())
    got["pointwise_l2_loss/nchw"] = L.pointwise_l2_loss(x["pred"], x["gt"], 0.01)
    got["pointwise_l2_loss/nhwc"] = L.pointwise_l2_loss(x["pred"].transpose(0, 2, 3, 1), x["gt"].transpose(0, 2, 3, 1), 0.01, data_format="NHWC")
    got["scale_invariant_gradient_loss"] = L.scale_invariant_gradient_loss(x["pred"], x["gt"], 0.01)
    got["compute_confidence_map"] = L.compute_confidence_map(x["flow_p"], x["flow_g"], scale=2)
    got["scale_invariant_gradient"] = L.scale_invariant_gradient(x["flow_p"], deltas=[1, 2, 4], weights=[1, 0.5, 0.25], epsilon=0.001)
    checked = 0
    for k in f.files:
        if k in ("backend", "l1_loss"):        # (l1_loss, v2/losses.py:23-29, has no caller in the reference and no mirror here)
            continue
        want, have = f[k], np.asarray(got[k], np.float32)
        assert have.shape == want.shape, (k, have.shape, want.shape)
        nan_w, nan_h = np.isnan(want), np.isnan(have)
        assert np.array_equal(nan_w, nan_h), "%s: NaN pattern differs at %d positions" % (k, int((nan_w != nan_h).sum()))
        ok = ~nan_w
        np.testing.assert_allclose(have[ok], want[ok], rtol=tol, atol=tol, err_msg=k)
        checked += 1
    assert checked == 15


def test_losses_mirror_is_wired_like_the_reference(monkeypatch):
    from demon_amd import losses
    monkeypatch.setattr(losses, "sops", _OracleSops())
    _check_losses(losses, 1e-6)


@pytest.mark.gpu
def test_losses_on_the_hip_ops_equal_the_reference_wiring():
    from demon_amd import losses
    _check_losses(losses, 2e-5)
