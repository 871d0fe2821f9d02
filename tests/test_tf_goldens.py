"""Consumers of the golden vectors that tools/dump_reference_goldens.py writes in the REFERENCE's environment (TensorFlow 1.4 +
lmbspecialops + the reference's depthmotionnet): tests/golden/tf/{manifest.json, ops.npz, layers.npz, nets_original.npz, nets_v2.npz,
ckpt/netRefine_seed1.*}.  With those files committed, every test below holds the CPU oracle (-m "not gpu") and the HIP path (-m gpu)
to the reference's own outputs -- the pin DESIGN.md section 4 lists as missing.  Without them the `reference` variants skip.

The `selftest` variants run the same checks on files the tool writes with this repo's oracle in TensorFlow's place
(`--self-test`): they pin nothing about the reference (oracle against oracle on the CPU; HIP against oracle on the GPU), but they
keep the file formats, the seeded-weight recipe shared with the tool (tools/golden_common.py) and every consumer alive.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_l1

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import golden_common as G  # noqa: E402
import dump_reference_goldens as D  # noqa: E402

REFERENCE_DIR = os.path.join(ROOT, "tests", "golden", "tf")
NET_TOL = 1e-3     # BASELINE.json north_star: outputs within 1e-3 relative L1 of the reference


@pytest.fixture(scope="session")
def selftest_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("tfgold"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dump_reference_goldens.py"), "--self-test", d],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return d


@pytest.fixture(params=["reference", "selftest"])
def golden(request):
    if request.param == "reference":
        if not os.path.isfile(os.path.join(REFERENCE_DIR, "manifest.json")):
            pytest.skip("tests/golden/tf/ is empty: run tools/dump_reference_goldens.py in the reference's environment")
        d = REFERENCE_DIR
    else:
        d = request.getfixturevalue("selftest_dir")
    with open(os.path.join(d, "manifest.json")) as f:
        manifest = json.load(f)
    assert manifest["format_version"] == G.FORMAT_VERSION
    assert (manifest["backend"] == "oracle-selftest") == (request.param == "selftest")   # a self-test dump must never sit in tests/golden/tf
    # ... and neither must a dry run of the tool on the TensorFlow emulator (oracle/tf1, tests/test_tf1_emulator.py)
    assert not str(manifest["versions"].get("tensorflow", "")).endswith("-emulated")
    return d, manifest


def _close(got, want, tag, rtol=1e-4, atol=1e-5, l1=None):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert got.shape == want.shape, "%s: shape %s vs golden %s" % (tag, got.shape, want.shape)
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), "%s: NaN pattern differs at %d positions" % (tag, int((gn != wn).sum()))
    gi, wi = np.isinf(got), np.isinf(want)
    assert np.array_equal(gi, wi) and np.array_equal(got[gi], want[wi]), "%s: infinities differ" % tag
    ok = ~(wn | wi)
    if l1 is not None:
        assert rel_l1(got[ok], want[ok]) <= l1, "%s: relative L1 %.3e" % (tag, rel_l1(got[ok], want[ok]))
    else:
        np.testing.assert_allclose(got[ok], want[ok], rtol=rtol, atol=atol, err_msg=tag)


# ---- implementations under test ----------------------------------------------------------------------------------------------------
class _OracleOps:
    def __getattr__(self, op):
        from oracle import ops_ref
        return getattr(ops_ref, op)


class _HipOps:
    def __getattr__(self, op):
        from demon_amd import sops
        return getattr(sops, op)


class _HipLayers:
    """the interface apply_layer expects (oracle/ops_ref.py) on the HIP layer entry points of the C ABI"""

    def __init__(self, ctx):
        self.ctx = ctx

    def conv2d_hwio(self, x, w, b, stride, pad, lrelu):
        assert tuple(pad) == (w.shape[0] // 2, w.shape[1] // 2)
        return self.ctx.conv2d(x, w, b, stride, lrelu, padding="caffe")

    def conv2d_hwio_same(self, x, w, b, stride, lrelu):
        return self.ctx.conv2d(x, w, b, stride, lrelu, padding="same")

    def deconv4x4s2_crop(self, x, w, b, lrelu):
        return self.ctx.deconv4x4s2(x, w, b, lrelu)

    def dense(self, x, w, b, lrelu):
        return self.ctx.dense(x, w, b, lrelu)


def _check_ops(d, impl, ill_conditioned_l1):
    with np.load(os.path.join(d, "ops.npz")) as f:
        cases = G.unpack_cases(f)
    fresh = G.op_cases()
    assert [c["tag"] for c in cases] == [c["tag"] for c in fresh]
    for c, mine in zip(cases, fresh):
        for (name, arr), (name2, arr2) in zip(c["inputs"], mine["inputs"]):      # the inputs both sides regenerate from the seed
            assert name == name2 and np.array_equal(arr, np.asarray(arr2, np.float32), equal_nan=True), c["tag"]
        got = getattr(impl, c["op"])(*[a for _, a in c["inputs"]], **c["kwargs"])
        # triangulating depth from flow that no depth explains is ill conditioned: those cases are held to the parity bar of the nets
        loose = c["op"].startswith("flow_to_depth") and not c["tag"].split("/")[1] in ("consistent", "zero")
        _close(got, c["out"], c["tag"], l1=ill_conditioned_l1 if loose else None)
    return len(cases)


def _check_layers(d, ops):
    with np.load(os.path.join(d, "layers.npz")) as f:
        cases = G.layer_cases()
        assert int(f["count"]) == len(cases)
        for i, c in enumerate(cases):
            assert str(f["%d/tag" % i]) == c["tag"]
            variables = [(n, tuple(s)) for n, s in json.loads(str(f["%d/variables" % i]))]
            assert sorted(variables) == sorted(D.G_layer_variables(c)), c["tag"]           # the names / shapes tf.layers really created
            w = G.seeded_weights(variables, D.SEED + 100, head_scale=1.0)
            _close(D.apply_layer(ops, c, w), f["%d/out" % i], c["tag"], l1=1e-5)
    return len(cases)


def _nets(d, version):
    path = os.path.join(d, "nets_original.npz" if version == 1 else "nets_v2.npz")
    if not os.path.isfile(path):
        pytest.skip("%s not dumped (the v2 graphs need a GPU build of TensorFlow)" % os.path.basename(path))
    return np.load(path)


def _check_net_stages(f, run):
    """run(stage, feeds) -> dict of outputs; stages chained on the GOLDEN outputs of the previous stage (each eval() is pinned on
    its own: a deviation in one stage cannot hide behind, or be blamed on, the next)"""
    worst = 0.0
    for case in json.loads(str(f["cases"])):
        pair, img2_2 = f["%s/in/image_pair" % case], f["%s/in/image2_2" % case]

        def gold(stage):
            pre = "%s/out/%s/" % (case, stage)
            return {k[len(pre):]: f[k] for k in f.files if k.startswith(pre)}

        prev = None
        for stage in ("bootstrap", "iterative0", "iterative1", "iterative2", "refine"):
            want = gold(stage)
            got = run(stage, pair, img2_2, prev)
            for k, v in want.items():
                err = rel_l1(got[k], v)
                worst = max(worst, err)
                assert np.isfinite(got[k]).all() and err <= NET_TOL, "%s %s %s: relative L1 %.3e" % (case, stage, k, err)
            prev = want
    return worst


# ---- CPU: the oracle against the goldens -------------------------------------------------------------------------------------------
def test_ops_oracle(golden):
    d, _ = golden
    assert _check_ops(d, _OracleOps(), NET_TOL) >= 60


def test_layers_oracle(golden):
    d, _ = golden
    from oracle import ops_ref
    assert _check_layers(d, ops_ref) >= 15


@pytest.mark.parametrize("version", [1, 2])
def test_variable_names_and_shapes(golden, version):
    """Appendix B of SURVEY.md (tf.layers naming rules) against what TensorFlow really created: the set demon_set_weight accepts"""
    d, _ = golden
    from demon_amd import weights as W
    f = _nets(d, version)
    theirs = sorted((n, tuple(s)) for n, s in json.loads(str(f["variables"])))
    ours = sorted((n, tuple(s)) for n, s in W.variable_shapes(version=version).items())
    assert theirs == ours


def test_checkpoint_written_by_tf_saver_is_read(golden):
    """demon_amd/tf_checkpoint.py on a bundle that tf.train.Saver wrote (reference dump) -- not on its own writer's output"""
    d, _ = golden
    from demon_amd import tf_checkpoint as ck
    from demon_amd import weights as W
    prefix = os.path.join(d, "ckpt", "netRefine_seed%d" % D.SEED)
    if not os.path.isfile(prefix + ".index"):
        pytest.skip("no checkpoint in this dump")
    shapes = {n: s for n, s in W.variable_shapes(version=1).items() if n.startswith("netRefine/")}
    want = G.seeded_weights(sorted(shapes.items()), D.SEED)
    index, _ = ck.read_index(prefix)
    assert sorted(index) == sorted(shapes)
    got = ck.load_tf_checkpoint(prefix, sorted(shapes))
    for n in shapes:
        assert got[n].dtype == np.float32 and got[n].shape == tuple(shapes[n])
        np.testing.assert_array_equal(got[n], want[n], err_msg=n)


@pytest.mark.parametrize("version", [1, 2])
def test_nets_oracle(golden, version):
    d, _ = golden
    from demon_amd import weights as W
    from oracle import net_ref
    f = _nets(d, version)
    w = G.seeded_weights(sorted(W.variable_shapes(version=version).items()), int(f["seed"]), consistent_flow=(version == 2))
    net = net_ref.DemonRef(w) if version == 1 else net_ref.DemonRefV2(w)

    def run(stage, pair, img2_2, prev):
        if stage == "bootstrap":
            return net.bootstrap(pair, img2_2)
        if stage == "refine":
            return net.refine(pair[:, 0:3], prev["predict_depth2"])
        return net.iterative(pair, img2_2, prev["predict_depth2"], prev["predict_normal2"], prev["predict_rotation"], prev["predict_translation"])

    _check_net_stages(f, run)


# ---- GPU: the HIP path against the goldens -----------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_ops_hip(golden):
    d, _ = golden
    assert _check_ops(d, _HipOps(), NET_TOL) >= 60


@pytest.mark.gpu
def test_layers_hip(golden, gpu_ctx):
    d, _ = golden
    assert _check_layers(d, _HipLayers(gpu_ctx)) >= 15


@pytest.mark.gpu
@pytest.mark.parametrize("version", [1, 2])
def test_nets_hip(golden, version):
    d, _ = golden
    from demon_amd import DemonContext, weights as W
    f = _nets(d, version)
    w = G.seeded_weights(sorted(W.variable_shapes(version=version).items()), int(f["seed"]), consistent_flow=(version == 2))
    ctx = DemonContext(0, 1, 192, 256, version=version)
    try:
        ctx.set_weights(w)

        def run(stage, pair, img2_2, prev):
            if stage == "bootstrap":
                return ctx.bootstrap(pair, img2_2)
            if stage == "refine":
                return ctx.refine(pair[:, 0:3], prev["predict_depth2"])
            return ctx.iterative(pair, img2_2, prev["predict_depth2"], prev["predict_normal2"], prev["predict_rotation"], prev["predict_translation"])

        _check_net_stages(f, run)
    finally:
        ctx.close()
