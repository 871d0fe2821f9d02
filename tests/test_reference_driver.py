"""The LITERAL reference driver -- the unmodified file examples/example.py of lmb-freiburg/demon -- on this repo's
`depthmotionnet` package and TensorFlow stand-in.

The script resolves everything relative to its own location (examples/example.py:8-10: `<examples>/../python` goes first on
sys.path, `<examples>/../weights/demon_original` is the checkpoint, the sculpture PNGs sit next to it), so the test builds that
layout in a temporary directory out of symlinks:

    <tmp>/examples/example.py      -> the reference file, untouched
    <tmp>/examples/sculpture?.png  -> the reference's fixtures (or the same pixels from tests/golden/sculpture_inputs.npz)
    <tmp>/python                   -> <repo>/python            (depthmotionnet = the MI355X-native mirror)
    <tmp>/weights/demon_original.* -> a TensorBundle checkpoint written by demon_amd.tf_checkpoint (synthetic weights: the
                                      real ones need the network, weights/download_weights.sh:2-3)

and runs it with PYTHONPATH=<repo>/python/tf_stub:<repo>, MPLBACKEND=Agg, DEMON_EVAL_DUMP=<tmp>/result.npz (the network classes
leave what they evaluated there at exit).  The reference file is looked up in $DEMON_REFERENCE, /root/reference, or
$DEMON_REFERENCE_EXAMPLES (a directory holding example.py); without it the tests skip -- it is not part of this repo and must
not be copied into it.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _reference_examples():
    cands = [os.environ.get("DEMON_REFERENCE_EXAMPLES")]
    for base in (os.environ.get("DEMON_REFERENCE"), "/root/reference"):
        if base:
            cands.append(os.path.join(base, "examples"))
    cands.append(os.path.join(ROOT, "gpurun_in", "reference_examples"))   # scratch staging for a one-off GPU run (git-ignored)
    for c in cands:
        if c and os.path.isfile(os.path.join(c, "example.py")):
            return c
    return None


def _stage(tmp, ref_examples, weights, script="example.py", checkpoint="demon_original"):
    from demon_amd import tf_checkpoint as ck
    ex = os.path.join(tmp, "examples")
    os.makedirs(ex)
    os.symlink(os.path.join(ref_examples, script), os.path.join(ex, script))
    g = np.load(os.path.join(ROOT, "tests", "golden", "sculpture_inputs.npz"))
    for i in (1, 2):
        src = os.path.join(ref_examples, "sculpture%d.png" % i)
        dst = os.path.join(ex, "sculpture%d.png" % i)
        if os.path.isfile(src):
            os.symlink(src, dst)
        else:   # the same 256x192 pixels the reference's prepare_input_data produced from its PNGs (make_golden_inputs.py)
            from PIL import Image
            Image.fromarray(g["image%d_u8" % i]).save(dst)
    os.symlink(os.path.join(ROOT, "python"), os.path.join(tmp, "python"))
    os.makedirs(os.path.join(tmp, "weights"))
    ck.save_tf_checkpoint(os.path.join(tmp, "weights", checkpoint), weights)
    return os.path.join(ex, script)


def _run(script, dump, args=()):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "python", "tf_stub"), ROOT])
    env["MPLBACKEND"] = "Agg"
    env["DEMON_EVAL_DUMP"] = dump
    env.pop("DEMON_SYNTHETIC_WEIGHTS", None)
    return subprocess.run([sys.executable, script] + list(args), capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(script))


@pytest.mark.gpu
def test_unmodified_reference_example_runs_and_equals_demon_full(tmp_path, synth_weights):
    ref = _reference_examples()
    if ref is None:
        pytest.skip("the reference's examples/example.py is not on this machine ($DEMON_REFERENCE)")
    script = _stage(str(tmp_path), ref, synth_weights)
    with open(script) as f:
        assert "from depthmotionnet.networks_original import *" in f.read()    # it is the reference driver, not examples/ of this repo
    dump = str(tmp_path / "result.npz")
    r = _run(script, dump)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    res = np.load(dump)
    assert str(res["data_format"]) == "channels_first"          # tf.test.is_gpu_available(True) answered by libdemon_hip.so
    assert int(res["BootstrapNet.calls"]) == 1 and int(res["IterativeNet.calls"]) == 3 and int(res["RefinementNet.calls"]) == 1
    from demon_amd import DemonContext
    ctx = DemonContext(0, 1, 192, 256)
    try:
        ctx.set_weights(synth_weights)
        ctx.load_tuned_plan(1)
        want = ctx.full(res["BootstrapNet.image_pair"], res["BootstrapNet.image2_2"], iterations=3)
    finally:
        ctx.close()
    np.testing.assert_array_equal(res["RefinementNet.predict_depth0"], want["predict_depth0"])
    np.testing.assert_array_equal(res["IterativeNet.predict_rotation"], want["predict_rotation"])
    np.testing.assert_array_equal(res["IterativeNet.predict_translation"], want["predict_translation"])
    np.testing.assert_array_equal(res["IterativeNet.predict_depth2"], want["predict_depth2"])
    # the images the script prepared are the reference's own prepare_input_data output (golden, Pillow's default filter)
    g = np.load(os.path.join(ROOT, "tests", "golden", "sculpture_inputs.npz"))
    np.testing.assert_array_equal(res["RefinementNet.image1"][0], (g["image1_u8"].astype(np.float32) / 255 - 0.5).transpose(2, 0, 1))


def test_unmodified_reference_example_reaches_the_gpu_boundary_on_cpu(tmp_path, synth_weights):
    """Without a GPU the same literal script must get through its imports, prepare_input_data on the PNGs and the TensorFlow
    stand-in (GPU probe -> channels_last, GPUOptions / ConfigProto / InteractiveSession) and then FAIL LOUDLY at the first
    statement that needs the device, `BootstrapNet(session, data_format)` (examples/example.py:75): there is no CPU fallback."""
    import torch
    ref = _reference_examples()
    if ref is None:
        pytest.skip("the reference's examples/example.py is not on this machine ($DEMON_REFERENCE)")
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu test")
    script = _stage(str(tmp_path), ref, synth_weights)
    r = _run(script, str(tmp_path / "result.npz"))
    assert r.returncode != 0
    tail = r.stderr[-3000:]
    assert "bootstrap_net = BootstrapNet(session, data_format)" in tail, tail
    assert "demon_create failed" in tail and "no HIP device" in tail, tail
    assert not os.path.exists(str(tmp_path / "result.npz"))     # nothing was evaluated


# ---- the literal v2 driver, examples/example_v2.py:50-105 (needs --checkpoint; exits with "requires a GPU" without one) ----------

def _reference_v2():
    ref = _reference_examples()
    return ref if ref and os.path.isfile(os.path.join(ref, "example_v2.py")) else None


@pytest.mark.gpu
def test_unmodified_reference_example_v2_runs_and_equals_demon_full():
    from demon_amd import weights as W
    import tempfile
    ref = _reference_v2()
    if ref is None:
        pytest.skip("the reference's examples/example_v2.py is not on this machine ($DEMON_REFERENCE)")
    w2 = W.synthetic_weights(seed=1, version=2)
    with tempfile.TemporaryDirectory() as tmp:
        script = _stage(tmp, ref, w2, script="example_v2.py", checkpoint="demon_v2")
        with open(script) as f:
            assert "from depthmotionnet.v2.networks import *" in f.read()
        dump = os.path.join(tmp, "result.npz")
        r = _run(script, dump, ["--checkpoint", os.path.join(tmp, "weights", "demon_v2")])
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        res = dict(np.load(dump))
    assert int(res["BootstrapNet.calls"]) == 1 and int(res["IterativeNet.calls"]) == 3 and int(res["RefinementNet.calls"]) == 1
    from demon_amd import DemonContext
    ctx = DemonContext(0, 1, 192, 256, version=2)
    try:
        ctx.set_weights(w2)
        ctx.load_tuned_plan(1)
        want = ctx.full(res["BootstrapNet.image_pair"], res["BootstrapNet.image2_2"], iterations=3)
    finally:
        ctx.close()
    np.testing.assert_array_equal(res["RefinementNet.predict_depth0"], want["predict_depth0"])
    np.testing.assert_array_equal(res["RefinementNet.predict_normal0"], want["predict_normal0"])
    np.testing.assert_array_equal(res["IterativeNet.predict_rotation"], want["predict_rotation"])
    np.testing.assert_array_equal(res["IterativeNet.predict_translation"], want["predict_translation"])
    np.testing.assert_array_equal(res["IterativeNet.predict_depth2"], want["predict_depth2"])


def test_unmodified_reference_example_v2_stops_without_a_gpu():
    """examples/example_v2.py:50-54: without a GPU the script prints 'Running this example requires a GPU' and exits 1 -- the
    stand-in's tf.test.is_gpu_available answers through libdemon_hip.so, so that is what happens here (no CPU fallback)."""
    import tempfile
    import torch
    from demon_amd import weights as W
    ref = _reference_v2()
    if ref is None:
        pytest.skip("the reference's examples/example_v2.py is not on this machine ($DEMON_REFERENCE)")
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu test")
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "examples"))
        os.symlink(os.path.join(ref, "example_v2.py"), os.path.join(tmp, "examples", "example_v2.py"))
        os.symlink(os.path.join(ROOT, "python"), os.path.join(tmp, "python"))
        r = _run(os.path.join(tmp, "examples", "example_v2.py"), os.path.join(tmp, "result.npz"), ["--checkpoint", os.path.join(tmp, "nothing")])
        assert r.returncode == 1 and "Running this example requires a GPU" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
        assert not os.path.exists(os.path.join(tmp, "result.npz"))
