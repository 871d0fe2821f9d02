"""GPU parity of the three networks and the device-resident full pipeline against the CPU oracle on the
same seeded inputs and weights, through the C ABI.  Gate: relative L1 <= 1e-3 per output tensor
(BASELINE.json north_star); typical observed values are ~1e-6..1e-5 (summation order only)."""
import os
import numpy as np
import pytest

from conftest import rel_l1, make_inputs
from oracle import net_ref

pytestmark = pytest.mark.gpu
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-3
KEYS = ("predict_flow5", "predict_flow2", "predict_depth2", "predict_normal2", "predict_rotation", "predict_translation")


@pytest.fixture(scope="module")
def ref(synth_weights):
    return net_ref.DemonRef(synth_weights)


def _cmp(got, want, keys, tol=TOL):
    for k in keys:
        assert got[k].shape == want[k].shape, k
        assert np.isfinite(got[k]).all(), k
        err = rel_l1(got[k], want[k])
        assert err < tol, "%s rel L1 %.3e" % (k, err)


def test_variable_table_matches_library(gpu_ctx):
    from demon_amd import weights
    lib_vars = dict(gpu_ctx.variables())
    assert lib_vars == weights.variable_shapes()
    assert gpu_ctx.variables() == weights.blob_order()   # same ORDER too: it is the layout of the flat weight blob
    assert gpu_ctx.blob_size() == 45753883


@pytest.mark.parametrize("n", [1, 3])
def test_bootstrap(gpu_ctx, ref, n):
    pair, img2_2 = make_inputs(n, seed=n)
    _cmp(gpu_ctx.bootstrap(pair, img2_2), ref.bootstrap(pair, img2_2), KEYS + ("predict_conf5", "predict_conf2", "predict_scale"))


@pytest.mark.parametrize("method", [0, 1])
def test_iterative(gpu_ctx, synth_weights, method):
    ref = net_ref.DemonRef(synth_weights, flow_to_depth_method=method)
    gpu_ctx.set_option("flow_to_depth_method", method)
    try:
        pair, img2_2 = make_inputs(2, seed=5)
        b = ref.bootstrap(pair, img2_2)
        args = (pair, img2_2, b["predict_depth2"], b["predict_normal2"], b["predict_rotation"], b["predict_translation"])
        _cmp(gpu_ctx.iterative(*args), ref.iterative(*args), KEYS)
    finally:
        gpu_ctx.set_option("flow_to_depth_method", 0)


def test_iterative_gate_and_nan_inputs(gpu_ctx, ref):
    """bad camera parameters / invalid depth: the NaN gate of blocks_original.py:163-168 keeps outputs finite"""
    pair, img2_2 = make_inputs(2, seed=6)
    depth2 = np.full((2, 1, 48, 64), 0.5, np.float32)
    depth2[0, 0, :10] = -1.0        # invalid inverse depth -> NaN flow -> gated to 0
    depth2[1, 0, 5, 5] = 0.0        # 1 / 0 = inf depth -> NaN flow -> gated to 0
    normal2 = np.zeros((2, 3, 48, 64), np.float32)
    rot = np.array([[0.0, 0.0, 0.0], [0.3, -0.2, 0.1]], np.float32)
    tr = np.array([[5.0, 0.0, 0.0], [0.1, 0.9, -0.2]], np.float32)  # huge translation -> |flow| >= 1 -> 0
    args = (pair, img2_2, depth2, normal2, rot, tr)
    got = gpu_ctx.iterative(*args)
    assert all(np.isfinite(got[k]).all() for k in KEYS)
    _cmp(got, ref.iterative(*args), KEYS)


def test_iterative_nan_input_propagates_like_the_oracle(gpu_ctx, ref):
    """A NaN in the fed depth2 is NOT absorbed by the gate: the gate only zeroes the flow computed from it
    (blocks_original.py:163-168) while depth2 itself is a conv input (:180), so NaN spreads through the net -- on both sides
    alike.  Sample 0 carries the NaN, sample 1 is clean and must be unaffected (pairs are independent)."""
    pair, img2_2 = make_inputs(2, seed=16)
    depth2 = np.full((2, 1, 48, 64), 0.5, np.float32)
    depth2[0, 0, 5, 5] = np.nan
    normal2 = np.zeros((2, 3, 48, 64), np.float32)
    rot = np.array([[0.02, -0.01, 0.03], [0.3, -0.2, 0.1]], np.float32)
    tr = np.array([[0.5, 0.1, 0.0], [0.1, 0.9, -0.2]], np.float32)
    args = (pair, img2_2, depth2, normal2, rot, tr)
    got, want = gpu_ctx.iterative(*args), ref.iterative(*args)
    for k in KEYS:
        assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), k
        assert np.isnan(want[k][0]).any(), k                         # the NaN reaches every output of its own sample
        assert np.isfinite(got[k][1]).all(), k                       # ... and none of the other sample
        assert rel_l1(got[k][1], want[k][1]) < 1e-3, k
    clean = depth2.copy()
    clean[0, 0, 5, 5] = 0.5
    alone = gpu_ctx.iterative(pair, img2_2, clean, normal2, rot, tr)
    for k in KEYS:
        np.testing.assert_array_equal(alone[k][1], got[k][1])        # bit-identical: no cross-sample leakage


def test_refine(gpu_ctx, ref):
    pair, _ = make_inputs(2, seed=7)
    rng = np.random.default_rng(8)
    depth2 = (0.2 + rng.random((2, 1, 48, 64))).astype(np.float32)
    image1 = np.ascontiguousarray(pair[:, :3])
    _cmp(gpu_ctx.refine(image1, depth2), ref.refine(image1, depth2), ("predict_depth0",))


def test_full_pipeline_device_resident(gpu_ctx, ref):
    """bootstrap + 3 x iterative + refine without host round trips == the staged oracle (example.py:87-99)"""
    pair, img2_2 = make_inputs(2, seed=9)
    want = ref.full(pair, img2_2, iterations=3)
    got = gpu_ctx.full(pair, img2_2, iterations=3)
    _cmp(got, want, KEYS + ("predict_depth0",))
    # hipGraph replay is deterministic and equals the eager launch sequence bit for bit
    again = gpu_ctx.full(pair, img2_2, iterations=3)
    gpu_ctx.set_option("hipgraph", 0)
    try:
        eager = gpu_ctx.full(pair, img2_2, iterations=3)
    finally:
        gpu_ctx.set_option("hipgraph", 1)
    for k in KEYS + ("predict_depth0",):
        np.testing.assert_array_equal(got[k], again[k])
        np.testing.assert_array_equal(got[k], eager[k])
    # staged host API (5 calls like example.py) == device-resident loop
    r = gpu_ctx.bootstrap(pair, img2_2)
    for _ in range(3):
        r = gpu_ctx.iterative(pair, img2_2, r["predict_depth2"], r["predict_normal2"], r["predict_rotation"], r["predict_translation"])
    d0 = gpu_ctx.refine(np.ascontiguousarray(pair[:, :3]), r["predict_depth2"])["predict_depth0"]
    np.testing.assert_array_equal(d0, got["predict_depth0"])
    # batch independence: pairs do not interact (no batch statistics anywhere)
    single = gpu_ctx.full(pair[1:2], img2_2[1:2], iterations=3)
    assert rel_l1(single["predict_depth0"], got["predict_depth0"][1:2]) < 1e-5


def test_gate_stress_weights(gpu_ctx):
    """un-scaled heads: flows mostly >= 1 so the gate / NaN path dominates (SURVEY 8d 'gate-stress')"""
    from demon_amd import weights
    w = weights.synthetic_weights(seed=2, head_scale=1.0)
    gpu_ctx.set_weights(w)
    try:
        ref = net_ref.DemonRef(w)
        pair, img2_2 = make_inputs(1, seed=10)
        want = ref.full(pair, img2_2, iterations=1)
        got = gpu_ctx.full(pair, img2_2, iterations=1)
        # discontinuities (gate, floor) make per-pixel error chaotic: judge by aggregate relative L1
        _cmp(got, want, KEYS + ("predict_depth0",), tol=5e-3)
    finally:
        gpu_ctx.set_weights(weights.synthetic_weights(seed=1))


def test_reference_api_mirror_both_data_formats(synth_weights, ref):
    """depthmotionnet.networks_original drop-in: same classes / eval signatures / keys / shapes (example.py:75-99)"""
    import demon_amd
    from demon_amd.networks_original import BootstrapNet, IterativeNet, RefinementNet
    demon_amd.set_default_weights(synth_weights)
    pair, img2_2 = make_inputs(1, seed=11)
    results = {}
    for df in ("channels_first", "channels_last"):
        tr = (lambda a: a) if df == "channels_first" else (lambda a: np.ascontiguousarray(a.transpose(0, 2, 3, 1)))
        boot, it, rf = BootstrapNet(None, df), IterativeNet(None, df), RefinementNet(None, df)
        r = boot.eval(tr(pair), tr(img2_2))
        assert sorted(r) == sorted(KEYS)
        for _ in range(3):
            r = it.eval(tr(pair), tr(img2_2), r["predict_depth2"], r["predict_normal2"], r["predict_rotation"], r["predict_translation"])
        d0 = rf.eval(tr(pair[:, :3]), r["predict_depth2"])["predict_depth0"]
        assert d0.shape == ((1, 1, 192, 256) if df == "channels_first" else (1, 192, 256, 1))
        assert r["predict_flow2"].shape == ((1, 2, 48, 64) if df == "channels_first" else (1, 48, 64, 2))
        results[df] = (d0, r)
        with pytest.raises(ValueError):
            boot.eval(tr(pair)[:, :-1], tr(img2_2))
    np.testing.assert_array_equal(results["channels_first"][0], results["channels_last"][0].transpose(0, 3, 1, 2))
    want = ref.full(pair, img2_2, 3)
    assert rel_l1(results["channels_first"][0], want["predict_depth0"]) < TOL


def test_other_resolution_config5_geometry(synth_weights):
    """H x W generality needed by BASELINE config 5 (640x480), checked at a small multiple of 32"""
    from demon_amd import DemonContext, weights
    H, W = 96, 160
    w = weights.synthetic_weights(seed=3, height=H, width=W)
    ctx = DemonContext(0, 2, H, W)
    try:
        ctx.set_weights(w)
        pair, img2_2 = make_inputs(2, H, W, seed=12)
        want = net_ref.DemonRef(w).full(pair, img2_2, iterations=2)
        got = ctx.full(pair, img2_2, iterations=2)
        _cmp(got, want, KEYS + ("predict_depth0",))
    finally:
        ctx.close()


def test_error_reporting(gpu_ctx):
    from demon_amd import DemonError
    pair, img2_2 = make_inputs(5)
    with pytest.raises(DemonError):
        gpu_ctx.bootstrap(pair, img2_2)  # batch 5 > max_batch 4
    with pytest.raises(DemonError):
        gpu_ctx.set_option("no_such_option", 1)
    with pytest.raises(DemonError):
        gpu_ctx.set_weights({"netFlow1/conv1y/kernel": np.zeros((9, 1, 6, 32), np.float32)})  # incomplete


def test_autotune_keeps_results(gpu_ctx, ref):
    """demon_autotune only changes launch plans (kernel family / tile / split-K): outputs stay within tolerance"""
    pair, img2_2 = make_inputs(2, seed=13)
    before = gpu_ctx.full(pair, img2_2, iterations=2)
    gpu_ctx.autotune(2)
    after = gpu_ctx.full(pair, img2_2, iterations=2)
    want = ref.full(pair, img2_2, iterations=2)
    _cmp(after, want, KEYS + ("predict_depth0",))
    for k in KEYS + ("predict_depth0",):
        assert rel_l1(after[k], before[k]) < 1e-4, k


def test_example_script_runs_on_an_image_pair(tmp_path, synth_weights):
    """examples/example.py (the reference driver's flow) end to end: PNGs -> depth / motion, weights from a TensorBundle
    checkpoint written in the reference's own format"""
    import subprocess
    import sys
    from PIL import Image
    from demon_amd import tf_checkpoint as ck
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    rng = np.random.default_rng(40)
    base = rng.integers(0, 255, (192 + 8, 256 + 8, 3), dtype=np.uint8)
    Image.fromarray(base[:192, :256]).save(tmp_path / "a.png")
    Image.fromarray(base[4:196, 6:262]).save(tmp_path / "b.png")
    prefix = str(tmp_path / "weights" / "demon_original")
    ck.save_tf_checkpoint(prefix, synth_weights)
    out = str(tmp_path / "result.npz")
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "example.py"), str(tmp_path / "a.png"), str(tmp_path / "b.png"),
                        "--weights", prefix, "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = np.load(out)
    assert res["predict_depth0"].shape == (1, 1, 192, 256) and np.isfinite(res["predict_depth0"]).all()
    assert res["rotation"].shape == (1, 3) and res["translation"].shape == (1, 3)


def test_reuse_image_features_option_is_exact(gpu_ctx):
    """opt-in loop-invariant reuse (conv1 / conv2 of the iterative nets computed once per forward): bit-identical outputs"""
    pair, img2_2 = make_inputs(2, seed=14)
    base = gpu_ctx.full(pair, img2_2, iterations=3)
    gpu_ctx.set_option("reuse_image_features", 1)
    try:
        fast = gpu_ctx.full(pair, img2_2, iterations=3)
        one = gpu_ctx.full(pair, img2_2, iterations=1)
    finally:
        gpu_ctx.set_option("reuse_image_features", 0)
    for k in KEYS + ("predict_depth0",):
        np.testing.assert_array_equal(fast[k], base[k])
    assert np.isfinite(one["predict_depth0"]).all()


def test_side_branches_are_exact(gpu_ctx):
    """second-stream side branches (motion head, level-5 flow head, extra-input chains; fork / join by events, captured into
    the hipGraph): same kernels on the same data -> bit-identical outputs with the option on / off, graph / eager, and
    together with reuse_image_features; staged calls (bootstrap / iterative / refine) use them too"""
    pair, img2_2 = make_inputs(3, seed=15)
    results = {}
    try:
        for side in (1, 0):
            for graph in (1, 0):
                for reuse in (0, 1):
                    gpu_ctx.set_option("side_branches", side)
                    gpu_ctx.set_option("hipgraph", graph)
                    gpu_ctx.set_option("reuse_image_features", reuse)
                    results[(side, graph, reuse)] = gpu_ctx.full(pair, img2_2, iterations=3)
        gpu_ctx.set_option("side_branches", 1); gpu_ctx.set_option("hipgraph", 1); gpu_ctx.set_option("reuse_image_features", 0)
        r = gpu_ctx.bootstrap(pair, img2_2)
        for _ in range(3):
            r = gpu_ctx.iterative(pair, img2_2, r["predict_depth2"], r["predict_normal2"], r["predict_rotation"], r["predict_translation"])
    finally:
        gpu_ctx.set_option("side_branches", 1)
        gpu_ctx.set_option("hipgraph", 1)
        gpu_ctx.set_option("reuse_image_features", 0)
    ref = results[(0, 0, 0)]
    for key, got in results.items():
        for k in KEYS + ("predict_depth0",):
            np.testing.assert_array_equal(got[k], ref[k], err_msg="%s %s" % (key, k))
    for k in KEYS:
        np.testing.assert_array_equal(r[k], ref[k])


def test_two_context_pipeline_matches_single_context(gpu_ctx, synth_weights):
    """demon_amd.pipeline.Pipeline (asynchronous copies on page-locked host arrays, two contexts / streams so that the copies of
    one batch run under the kernels of another) returns what DemonContext.full returns, batch by batch"""
    from demon_amd.pipeline import Pipeline
    pair, img2_2 = make_inputs(12, seed=16)
    pipe = Pipeline(synth_weights, batch=4)
    try:
        got = pipe.run(pair, img2_2, iterations=2)
        again = pipe.run(pair, img2_2, iterations=2)
    finally:
        pipe.close()
    from demon_amd import DemonContext
    single = DemonContext(0, 4, 192, 256)     # same launch plan as the pipeline's contexts -> same summation order, same bits
    try:
        single.set_weights(synth_weights)
        single.load_tuned_plan(4, lanes=3)    # (a lane of a group loads the throughput-mode plan of the nearest batch size)
        for i in range(3):
            want = single.full(pair[4 * i:4 * i + 4], img2_2[4 * i:4 * i + 4], iterations=2)
            for k in KEYS + ("predict_depth0", "predict_scale"):
                np.testing.assert_array_equal(got[k][4 * i:4 * i + 4], want[k], err_msg=k)
            other = gpu_ctx.full(pair[4 * i:4 * i + 4], img2_2[4 * i:4 * i + 4], iterations=2)   # heuristic plan: other order
            for k in KEYS + ("predict_depth0",):
                assert rel_l1(other[k], want[k]) < 1e-4, k
    finally:
        single.close()
    for k in got:
        np.testing.assert_array_equal(got[k], again[k])


def test_fused_pairs_option_matches(gpu_ctx, ref):
    """fused k x 1 / 1 x k pair (conv_pair.hip: conv1 as one launch, intermediate in LDS) vs the two separate launches:
    same arithmetic in another summation order -> oracle parity holds and the two paths agree to 1e-5"""
    pair, img2_2 = make_inputs(3, seed=17)
    fused = gpu_ctx.full(pair, img2_2, iterations=2)   # default: on
    gpu_ctx.set_option("fused_pairs", 0)
    try:
        base = gpu_ctx.full(pair, img2_2, iterations=2)
    finally:
        gpu_ctx.set_option("fused_pairs", 1)
    _cmp(fused, ref.full(pair, img2_2, iterations=2), KEYS + ("predict_depth0",))
    for k in KEYS + ("predict_depth0",):
        assert rel_l1(fused[k], base[k]) < 1e-5, k
    assert any(not np.array_equal(fused[k], base[k]) for k in KEYS)   # the option really switched kernels


def test_fused_pairs_all_types_in_subprocess(tmp_path):
    """every instantiation of the fused pair kernel (9 / 7 / 3 taps, 32 and 64 channels; DEMON_FUSED_PAIRS_ALL is read once per
    process) against the default path, for the original and the v2 model"""
    import subprocess
    import sys
    script = tmp_path / "fused_all.py"
    script.write_text(
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import make_inputs, rel_l1\n"
        "from demon_amd import DemonContext, weights\n"
        "for version in (1, 2):\n"
        "    ctx = DemonContext(0, 2, version=version); ctx.set_weights(weights.synthetic_weights(1, version=version))\n"
        "    pair, img = make_inputs(2, seed=18)\n"
        "    fused = ctx.full(pair, img, 2)\n"
        "    ctx.set_option('fused_pairs', 0)\n"
        "    base = ctx.full(pair, img, 2)\n"
        "    for k in base:\n"
        "        assert np.isfinite(fused[k]).all() and rel_l1(fused[k], base[k]) < 1e-5, (version, k, rel_l1(fused[k], base[k]))\n"
        "    ctx.close()\n"
        "print('fused ok')\n" % (ROOT_DIR, os.path.join(ROOT_DIR, "tests")))
    env = dict(os.environ, DEMON_FUSED_PAIRS_ALL="1")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fused ok" in out.stdout, out.stdout + out.stderr


def test_fused_input_assembly_is_exact(gpu_ctx):
    """one-launch assembly of the extra inputs (depth_to_flow + gate + warp2d + concat; warp2d + concat + flow_to_depth; image +
    upsampled depth, blocks_original.py:155-183, :335-362, :475-482) == the chain of stand-alone op launches, bit for bit, with
    both flow_to_depth formulations, graph and eager, side branches on and off; gate-stress weights (NaN / zeroed flows) too"""
    from demon_amd import weights
    pair, img2_2 = make_inputs(3, seed=21)
    results = {}
    try:
        for method in (0, 1):
            gpu_ctx.set_option("flow_to_depth_method", method)
            for fused in (1, 0):
                for side, graph in ((1, 1), (0, 0)):
                    gpu_ctx.set_option("fused_inputs", fused)
                    gpu_ctx.set_option("side_branches", side)
                    gpu_ctx.set_option("hipgraph", graph)
                    results[(method, fused, side, graph)] = gpu_ctx.full(pair, img2_2, iterations=3)
    finally:
        for k, v in (("fused_inputs", 1), ("side_branches", 1), ("hipgraph", 1), ("flow_to_depth_method", 0)):
            gpu_ctx.set_option(k, v)
    for method in (0, 1):
        ref = results[(method, 0, 0, 0)]
        for key, got in results.items():
            if key[0] != method:
                continue
            for k in KEYS + ("predict_depth0",):
                np.testing.assert_array_equal(got[k], ref[k], err_msg="%s %s" % (key, k))
    assert not np.array_equal(results[(0, 1, 1, 1)]["predict_depth2"], results[(1, 1, 1, 1)]["predict_depth2"])   # the option is live
    from demon_amd import DemonContext
    ctx = DemonContext(0, 2, 192, 256)
    try:
        ctx.set_weights(weights.synthetic_weights(seed=2, head_scale=1.0))    # gate stress: large flows, invalid depths
        a = ctx.full(pair[:2], img2_2[:2], iterations=2)
        ctx.set_option("fused_inputs", 0)
        b = ctx.full(pair[:2], img2_2[:2], iterations=2)
        for k in KEYS + ("predict_depth0",):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    finally:
        ctx.close()


def test_reference_driver_statements_run_on_the_tensorflow_stub(tmp_path, synth_weights, gpu_ctx):
    """The statements of the reference's examples/example.py:45-99 (GPU probe, session, the three nets built BEFORE the
    checkpoint is restored, global_variables_initializer, Saver.restore, bootstrap + 3 x iterative + refinement) executed
    against python/tf_stub + python/depthmotionnet in a fresh interpreter, on the sculpture pair prepared by the reference's own
    prepare_input_data (golden arrays), weights from a TensorBundle checkpoint.  The unmodified reference file itself cannot be
    run by a test (it is not in this repo, needs matplotlib and the real weights); this is its call sequence, line for line.
    The result equals the device-resident pipeline of the C ABI on the same inputs, and two sessions keep their own weights."""
    import subprocess
    import sys
    from demon_amd import tf_checkpoint as ck, weights
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    prefix = str(tmp_path / "weights" / "demon_original")
    ck.save_tf_checkpoint(prefix, synth_weights)
    other = weights.synthetic_weights(seed=9)
    prefix2 = str(tmp_path / "weights2" / "demon_original")
    ck.save_tf_checkpoint(prefix2, other)
    out = str(tmp_path / "result.npz")
    code = r'''
import os, sys
import numpy as np
ROOT, PREFIX, PREFIX2, OUT = sys.argv[1:5]
sys.path.insert(0, os.path.join(ROOT, "python", "tf_stub"))
sys.path.insert(0, os.path.join(ROOT, "python"))
sys.path.insert(0, ROOT)
import tensorflow as tf
from depthmotionnet.networks_original import *

g = np.load(os.path.join(ROOT, "tests", "golden", "sculpture_inputs.npz"))
if tf.test.is_gpu_available(True):
    data_format = 'channels_first'
else:
    data_format = 'channels_last'
input_data = {k: g["%s_%s_nearest" % (k, data_format)] if k == "image2_2" else None for k in ("image2_2",)}
img1 = (g["image1_u8"].astype(np.float32) / 255 - 0.5)
img2 = (g["image2_u8"].astype(np.float32) / 255 - 0.5)
if data_format == 'channels_first':
    img1, img2 = img1.transpose(2, 0, 1), img2.transpose(2, 0, 1)
    input_data['image_pair'] = np.concatenate((img1, img2), axis=0)[np.newaxis]
else:
    input_data['image_pair'] = np.concatenate((img1, img2), axis=-1)[np.newaxis]
input_data['image1'] = img1[np.newaxis]

gpu_options = tf.GPUOptions()
gpu_options.per_process_gpu_memory_fraction = 0.8
session = tf.InteractiveSession(config=tf.ConfigProto(allow_soft_placement=True, gpu_options=gpu_options))
bootstrap_net = BootstrapNet(session, data_format)
iterative_net = IterativeNet(session, data_format)
refine_net = RefinementNet(session, data_format)
session.run(tf.global_variables_initializer())
saver = tf.train.Saver()
saver.restore(session, PREFIX)

# a second session with other weights must not disturb the first one's nets
session2 = tf.InteractiveSession()
bootstrap2 = BootstrapNet(session2, data_format)
tf.train.Saver().restore(session2, PREFIX2)

result = bootstrap_net.eval(input_data['image_pair'], input_data['image2_2'])
first = {k: v.copy() for k, v in result.items()}
for i in range(3):
    result = iterative_net.eval(input_data['image_pair'], input_data['image2_2'], result['predict_depth2'], result['predict_normal2'],
                                result['predict_rotation'], result['predict_translation'])
rotation = result['predict_rotation']
translation = result['predict_translation']
result = refine_net.eval(input_data['image1'], result['predict_depth2'])
r2 = bootstrap2.eval(input_data['image_pair'], input_data['image2_2'])
np.savez(OUT, depth0=result['predict_depth0'], rotation=rotation, translation=translation, boot_depth2=first['predict_depth2'],
         other_depth2=r2['predict_depth2'], image_pair=input_data['image_pair'], image2_2=input_data['image2_2'])
'''
    r = subprocess.run([sys.executable, "-c", code, root, prefix, prefix2, out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = np.load(out)
    from demon_amd import DemonContext
    single = DemonContext(0, 1, 192, 256)      # as the nets of the driver: batch 1 with the shipped batch-1 launch plan
    try:
        single.set_weights(synth_weights)
        single.load_tuned_plan(1)
        want = single.full(res["image_pair"], res["image2_2"], iterations=3)
    finally:
        single.close()
    np.testing.assert_array_equal(res["depth0"], want["predict_depth0"])
    np.testing.assert_array_equal(res["rotation"], want["predict_rotation"])
    np.testing.assert_array_equal(res["translation"], want["predict_translation"])
    assert np.isfinite(res["depth0"]).all() and res["depth0"].shape == (1, 1, 192, 256)
    assert not np.array_equal(res["other_depth2"], res["boot_depth2"])    # session2's nets ran on session2's weights


def test_chained_pairs_equal_the_two_launches(synth_weights):
    """conv2_1 / conv3_1 / conv4_1: k x 1 and 1 x k conv as ONE chained launch (plan kinds 6 / 7: conv_frag_chain_kernel /
    conv_stream_chain_kernel, a workgroup holds all channels of whole rows) against the two launches of the same kernel variant:
    bit for bit, for every variant that fits, and the profile shows the chained kernel really ran"""
    from demon_amd import DemonContext
    n = 3
    pair, img2_2 = make_inputs(n, seed=23)
    ctx = DemonContext(device=0, max_batch=n, height=192, width=256)
    try:
        ctx.set_weights(synth_weights)
        cases = {"conv2_1": [(5, 1), (5, 3), (5, 5), (5, 8), (5, 11), (4, 3), (4, 7)],
                 "conv3_1": [(5, 0), (5, 2), (5, 6), (5, 9), (5, 10), (4, 0), (4, 1), (4, 6), (4, 8)],
                 "conv4_1": [(5, 4), (4, 9)]}
        for pairname, variants in cases.items():
            for kind, v in variants:
                names = ["%s/%s" % (net, pairname) for net in ("netFlow1", "netDM1")]
                ctx.set_plan(n, {nm + ax: [kind, v, 1] for nm in names for ax in "yx"})
                two = ctx.bootstrap(pair, img2_2)
                ctx.set_plan(n, {nm + "y": [kind + 1 if kind == 5 else 7, v, 1] for nm in names})
                one = ctx.bootstrap(pair, img2_2)
                for k in two:
                    np.testing.assert_array_equal(one[k], two[k], err_msg="%s kind %d variant %d: %s" % (pairname, kind, v, k))
                tags = {r["name"]: r["kernel"] for r in ctx.profile_full(n, iterations=0, repeats=1)}
                assert "chain" in tags[names[0] + "y+x"], tags.get(names[0] + "y+x")
                assert names[0] + "y" not in tags
        # a variant whose tile does not hold all channels of whole rows is refused
        with pytest.raises(Exception):
            ctx.set_plan(n, {"netFlow1/conv2_1y": [6, 6, 1]})
    finally:
        ctx.close()


def test_lane_group_calibration_keeps_results(synth_weights):
    """demon_amd/lanes.py: several contexts on one GPU fed round robin.  Calibration measures lane counts and stream -> hardware-queue
    mappings (demon_release_streams / demon_acquire_streams behind placeholder streams) and must not change a single bit of what
    a lane computes; every lane equals a plain context with the same launch plan; lanes > 1 run without side branches."""
    from demon_amd import DemonContext
    from demon_amd.lanes import LaneGroup
    n = 2
    group = LaneGroup(synth_weights, lanes=3, batch=n)
    try:
        batches = [make_inputs(n, seed=40 + i) for i in range(3)]
        group.upload_inputs(batches)
        group.run_resident(n, 3, iterations=2)
        group.synchronize()
        before = [c.download_outputs(n) for c in group.ctxs]
        os.environ["DEMON_LANES_FULL_SWEEP"] = "1"     # (round 6: the sweep stops on a plateau unless told otherwise; here every cell is wanted)
        try:
            rates = group.calibrate(n, iterations=2, steps_per_lane=2, pads=(0, 1, 2))
        finally:
            os.environ.pop("DEMON_LANES_FULL_SWEEP", None)
        assert set(rates) >= {"1@0", "2@0", "3@0", "2@1", "3@2"} and all(v > 0 for v in rates.values())
        assert group.mapping["lanes"] == len(group) and 1 <= len(group) <= 3 and group.mapping["placeholder_streams"] in (0, 1, 2)
        group.run_resident(n, len(group), iterations=2)
        group.synchronize()
        for c, b in zip(group.ctxs, before):
            after = c.download_outputs(n)
            for k in after:
                np.testing.assert_array_equal(after[k], b[k], err_msg=k)
        single = DemonContext(0, n, 192, 256)
        try:
            single.set_weights(synth_weights)
            single.set_plan(n, group.ctxs[0].get_plan(n))
            want = single.full(*batches[0], iterations=2)
        finally:
            single.close()
        for k in before[0]:
            np.testing.assert_array_equal(before[0][k], want[k], err_msg=k)
    finally:
        group.close()


def test_released_streams_refuse_work(synth_weights):
    """demon_release_streams / demon_acquire_streams (include/demon_hip.h): between the two a context has no HIP stream; every entry
    point that would enqueue work must refuse (DEMON_ERR_NOT_READY) instead of falling onto the null stream, and after acquiring
    new streams the context computes what it computed before (its hipGraph is captured again on the new streams: a graph captured across the two streams of a context does not survive them)."""
    from demon_amd import DemonContext
    from demon_amd.engine import DemonError
    n = 1
    ctx = DemonContext(0, n, 192, 256)
    try:
        ctx.set_weights(synth_weights)
        pair, img2_2 = make_inputs(n, seed=77)
        want = ctx.full(pair, img2_2, iterations=1)
        ctx.release_streams()
        ctx.synchronize()                                   # nothing in flight: fine
        for call in (lambda: ctx.run_full(n, 1), lambda: ctx.upload_inputs(pair, img2_2), lambda: ctx.download_outputs(n),
                     lambda: ctx.leaky_relu(np.ones((4,), np.float32))):
            with pytest.raises(DemonError, match="demon_acquire_streams"):
                call()
        ctx.acquire_streams()
        got = ctx.full(pair, img2_2, iterations=1)
        for k in want:
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    finally:
        ctx.close()


def test_borrowed_context_works_after_the_lane_group_is_closed(synth_weights):
    """A context that ran with side branches (a graph captured across its two streams), became lane 0 of a group (streams exchanged by
    the calibration, side branches off) and got back on its own must compute what it computed before -- round 5: launching the graph
    exec captured before the exchange aborted inside the HIP runtime; demon_release_streams now drops the cached execs."""
    from demon_amd import DemonContext
    from demon_amd.lanes import LaneGroup
    n = 2
    ctx = DemonContext(0, n, 192, 256)
    try:
        ctx.set_weights(synth_weights)
        pair, img2_2 = make_inputs(n, seed=81)
        want = ctx.full(pair, img2_2, iterations=2)                 # side branches on: graph across both streams
        group = LaneGroup(first=ctx, lanes=3, batch=n)
        try:
            for c in group.ctxs:
                c.upload_inputs(pair, img2_2)
            mid = ctx.full(pair, img2_2, iterations=2)              # side branches off inside the group
            group.calibrate(n, iterations=2, steps_per_lane=2, pads=(0, 1))
        finally:
            group.close()
        assert ctx.get_option("side_branches") == 1
        got = ctx.full(pair, img2_2, iterations=2)                  # the pre-group graph key again
        for k in want:
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
            np.testing.assert_array_equal(mid[k], want[k], err_msg=k)
    finally:
        ctx.close()


def test_lane_calibration_winner_is_remembered_per_environment(synth_weights, tmp_path, monkeypatch):
    """LaneGroup.calibrate(reuse=True): the (lanes, placeholder streams) winner of one group is re-applied by the next group of the same
    process environment (mapping_key) through demon_lanes_apply, without measuring -- also across processes through $DEMON_LANES_CACHE --
    and the lanes compute what an uncalibrated context computes"""
    import json
    from demon_amd import DemonContext
    from demon_amd.lanes import LaneGroup
    n = 2
    cache = tmp_path / "lanes.json"
    monkeypatch.setenv("DEMON_LANES_CACHE", str(cache))
    monkeypatch.setattr(LaneGroup, "_cache", {})
    batches = [make_inputs(n, seed=50 + i) for i in range(3)]
    g1 = LaneGroup(synth_weights, lanes=3, batch=n)
    try:
        g1.upload_inputs(batches)
        rates = g1.calibrate(n, iterations=1, steps_per_lane=2, pads=(0, 1), reuse=True)     # nothing remembered yet: measures
        assert rates and "reused" not in g1.mapping
        won = dict(g1.mapping)
        key = g1.mapping_key()
    finally:
        g1.close()
    assert json.loads(cache.read_text())[key]["lanes"] == won["lanes"]
    monkeypatch.setattr(LaneGroup, "_cache", {})                                              # "another process": only the file is left
    g2 = LaneGroup(synth_weights, lanes=3, batch=n)
    try:
        g2.upload_inputs(batches)
        rates2 = g2.calibrate(n, iterations=1, steps_per_lane=2, pads=(0, 1), reuse=True)
        if rates2 == {}:                   # the remembered winner measured within 2.5 % of its remembered rate: taken without a sweep
            assert g2.mapping["reused"] is True and g2.mapping["lanes"] == won["lanes"] == len(g2)
            assert g2.mapping["placeholder_streams"] == won["placeholder_streams"]
        else:                              # (tiny batches are noisy: the full calibration ran instead)
            assert "reused" not in g2.mapping and g2.mapping["lanes"] == len(g2)
        assert g2.mapping["verified_pairs_per_s"] > 0 and g2.mapping["attempts"] >= 1
        won = dict(g2.mapping)
        g2.run_resident(n, 2 * len(g2), iterations=1)
        g2.synchronize()
        got = g2.ctxs[0].download_outputs(n)
    finally:
        g2.close()
    plain = DemonContext(0, n, 192, 256)
    try:
        plain.set_weights(synth_weights)
        plain.load_tuned_plan(n, lanes=3 if won["lanes"] > 1 else 1)
        if won["lanes"] > 1:
            plain.set_option("side_branches", 0)
        want = plain.full(*batches[0], iterations=1)
    finally:
        plain.close()
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
