"""oracle/tf1: the TensorFlow-1.4 graph emulator that lets the reference's own graph-building code and the TensorFlow branch of
tools/dump_reference_goldens.py run in this container (test infrastructure; see its header).

  * its primitives (tf.layers.conv2d VALID / SAME, conv2d_transpose VALID / SAME, dense, resize_nearest_neighbor, pad, split,
    norm / where) against PyTorch-CPU: the emulator restates TensorFlow's documented semantics in numpy tap by tap, oracle/net_ref.py
    restates them with PyTorch's convolutions -- two independent restatements that must agree;
  * TensorFlow's naming rules as far as the reference relies on them (scope/layer/kernel, default layer names, duplicate = error,
    uninitialised read = error, wrong feed shape = error);
  * a DRY RUN of tools/dump_reference_goldens.py's TensorFlow branch (`TFBackend`, never executed before round 5): the whole tool
    under `python -W error -X dev` with the reference's python/ on the path, as a CPU build (channels_last graphs) and as a GPU
    build (channels_first graphs + the v2 nets), compared file by file with the `--self-test` dump (oracle in TensorFlow's place).
    Needs /root/reference (this container); skips elsewhere.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_l1

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("DEMON_REFERENCE", "/root/reference")
EMU = os.path.join(ROOT, "oracle", "tf1")


@pytest.fixture()
def tf(monkeypatch):
    monkeypatch.syspath_prepend(EMU)
    for m in [m for m in sys.modules if m == "tensorflow" or m.startswith("tensorflow.") or m == "lmbspecialops"]:
        monkeypatch.delitem(sys.modules, m)
    import tensorflow
    assert tensorflow.EMULATED
    yield tensorflow
    for m in [m for m in sys.modules if m == "tensorflow" or m.startswith("tensorflow.") or m == "lmbspecialops"]:
        del sys.modules[m]


def _run_layer(tf, build, x, weights):
    with tf.Graph().as_default(), tf.Session() as s:
        ph = tf.placeholder(tf.float32, x.shape)
        y = build(ph)
        names = [v.name for v in tf.global_variables()]
        for v in tf.global_variables():
            v.load(weights[v.name.split(":")[0].split("/")[-1]], s)
        return s.run(y, feed_dict={ph: x}), names


@pytest.mark.parametrize("k,s,padding", [((3, 3), (1, 1), "valid"), ((9, 1), (2, 1), "valid"), ((1, 5), (1, 2), "valid"), ((3, 3), (2, 2), "same"),
                                         ((5, 1), (2, 1), "same"), ((1, 7), (1, 2), "SAME"), ((4, 4), (2, 2), "same")])
def test_conv2d_matches_pytorch(tf, k, s, padding):
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 5, 13, 18)).astype(np.float32)
    w = {"kernel": rng.standard_normal(k + (5, 7)).astype(np.float32), "bias": rng.standard_normal(7).astype(np.float32)}
    want_in = torch.from_numpy(x)
    if padding.lower() == "same":     # TensorFlow: out = ceil(in / s), the odd padding element goes to the END
        pads = []
        for size, kk, ss in ((x.shape[3], k[1], s[1]), (x.shape[2], k[0], s[0])):
            total = max((-(-size // ss) - 1) * ss + kk - size, 0)
            pads += [total // 2, total - total // 2]
        want_in = F.pad(want_in, pads)
    want = F.conv2d(want_in, torch.from_numpy(w["kernel"].transpose(3, 2, 0, 1).copy()), torch.from_numpy(w["bias"]), stride=s).numpy()
    for df in ("channels_first", "channels_last"):
        xin = x if df == "channels_first" else x.transpose(0, 2, 3, 1)
        got, names = _run_layer(tf, lambda t: tf.layers.conv2d(t, 7, k, strides=s, padding=padding, data_format=df, name="c"), xin, w)
        got = got if df == "channels_first" else got.transpose(0, 3, 1, 2)
        assert names == ["c/kernel:0", "c/bias:0"]
        assert got.shape == want.shape and rel_l1(got, want) < 1e-5, (df, rel_l1(got, want))


@pytest.mark.parametrize("padding", ["VALID", "same"])
def test_conv2d_transpose_matches_pytorch(tf, padding):
    """blocks_original.py:97-110 ('VALID', then a crop by one) and :64-75 ('same'): 2H + 2 resp. 2H outputs; the 'same' result is the
    VALID one without its first and last row / column; kernel variable [4,4,Cout,Cin]"""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 6, 5, 7)).astype(np.float32)
    w = {"kernel": rng.standard_normal((4, 4, 3, 6)).astype(np.float32), "bias": rng.standard_normal(3).astype(np.float32)}
    full = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w["kernel"].transpose(3, 2, 0, 1).copy()), torch.from_numpy(w["bias"]), stride=2).numpy()
    want = full if padding == "VALID" else full[:, :, 1:-1, 1:-1]
    assert want.shape[2:] == ((12, 16) if padding == "VALID" else (10, 14))
    for df in ("channels_first", "channels_last"):
        xin = x if df == "channels_first" else x.transpose(0, 2, 3, 1)
        got, names = _run_layer(tf, lambda t: tf.layers.conv2d_transpose(t, 3, 4, strides=2, padding=padding, data_format=df, name="upconv"), xin, w)
        got = got if df == "channels_first" else got.transpose(0, 3, 1, 2)
        assert names == ["upconv/kernel:0", "upconv/bias:0"]
        assert got.shape == want.shape and rel_l1(got, want) < 1e-5, (df, padding)


def test_small_ops_and_naming_rules(tf):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 4, 6, 8)).astype(np.float32)
    with tf.Graph().as_default(), tf.Session() as s:
        ph = tf.placeholder(tf.float32, x.shape)
        a, b = tf.split(value=ph, num_or_size_splits=[1, 3], axis=1)
        assert a.get_shape().as_list() == [3, 1, 6, 8] and b.get_shape().as_list() == [3, 3, 6, 8]
        nrm = tf.norm(ph, axis=1, keep_dims=True)
        gated = tf.where(tf.concat((nrm, nrm, nrm, nrm), axis=1) < 2.0, ph, tf.zeros_like(ph, dtype=tf.float32))
        up = tf.image.resize_nearest_neighbor(tf.transpose(ph, [0, 2, 3, 1]), (24, 32))
        with tf.variable_scope("netX"):
            with tf.variable_scope("motion"):
                d = tf.layers.dense(name="fc", inputs=tf.contrib.layers.flatten(ph), units=5)
            c1 = tf.layers.conv2d(ph, 2, 1, data_format="channels_first")
            c2 = tf.layers.conv2d(ph, 2, 1, data_format="channels_first")
            with pytest.raises(ValueError, match="already exists"):
                tf.layers.conv2d(ph, 2, 1, data_format="channels_first", name="conv2d")
        names = [v.name for v in tf.global_variables()]
        assert names == ["netX/motion/fc/kernel:0", "netX/motion/fc/bias:0", "netX/conv2d/kernel:0", "netX/conv2d/bias:0",
                         "netX/conv2d_1/kernel:0", "netX/conv2d_1/bias:0"]
        assert [v.get_shape().as_list() for v in tf.global_variables()][:2] == [[192, 5], [5]]
        with pytest.raises(RuntimeError, match="uninitialized"):
            s.run(d, feed_dict={ph: x})
        with pytest.raises(ValueError, match="Cannot feed value of shape"):
            s.run(nrm, feed_dict={ph: x[:, :3]})
        with pytest.raises(RuntimeError, match="must feed"):
            s.run(nrm)
        with pytest.raises(TypeError):
            tf.norm(ph, axis=1, keepdims=True)          # TensorFlow 1.4 spells it keep_dims (blocks_original.py:165)
        assert not hasattr(tf, "keras") and not hasattr(tf, "compat")
        n, g, u, (ra, rb) = s.run([nrm, gated, up, (a, b)], feed_dict={ph: x})
        np.testing.assert_allclose(n, np.sqrt((x * x).sum(1, keepdims=True)), rtol=1e-6)
        np.testing.assert_array_equal(g, np.where(np.sqrt((x * x).sum(1, keepdims=True)) < 2.0, x, 0))
        np.testing.assert_array_equal(u, x.transpose(0, 2, 3, 1)[:, np.arange(24) // 4][:, :, np.arange(32) // 4])
        np.testing.assert_array_equal(np.concatenate([ra, rb], 1), x)
        assert c1.get_shape().as_list() == c2.get_shape().as_list() == [3, 2, 6, 8]


# ---- the dump tool's TensorFlow branch, executed ------------------------------------------------------------------------------------
def _dump(outdir, args, env_extra):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([EMU, ROOT])
    env.update(env_extra)
    return subprocess.run([sys.executable, "-W", "error", "-X", "dev", os.path.join(ROOT, "tools", "dump_reference_goldens.py")] + args + [outdir],
                          capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)


@pytest.fixture(scope="module")
def selftest_dump(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("selftest"))
    r = _dump(d, ["--self-test"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    return d


@pytest.mark.parametrize("gpu_build", [False, True])
def test_dump_tool_tensorflow_branch_dry_run(tmp_path, selftest_dump, gpu_build):
    if not os.path.isfile(os.path.join(REF, "python", "depthmotionnet", "networks_original.py")):
        pytest.skip("the reference checkout is not on this machine ($DEMON_REFERENCE)")
    d = str(tmp_path / "dump")
    r = _dump(d, ["--reference", REF, "--outdir"], {"TF1_EMU_GPU": "1" if gpu_build else "0"})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Warning" not in r.stderr, r.stderr[-3000:]           # -W error -X dev: nothing ignored either
    with open(os.path.join(d, "manifest.json")) as f:
        manifest = json.load(f)
    assert manifest["backend"] == "tensorflow+lmbspecialops" and manifest["versions"]["tensorflow"] == "1.4.0-emulated"
    assert manifest["versions"]["data_format"] == ("channels_first" if gpu_build else "channels_last")
    files = set(manifest["files"])
    assert {"ops.npz", "layers.npz", "nets_original.npz", "ckpt/netRefine_seed1.index"} <= files
    assert ("nets_v2.npz" in files) == gpu_build                  # v2/networks.py builds channels_first graphs only
    for name in sorted(n for n in files if n.endswith(".npz")):
        a, b = np.load(os.path.join(d, name)), np.load(os.path.join(selftest_dump, name))
        extra = set(a.files) - set(b.files)
        assert all(k.startswith("sculpture/") for k in extra) and set(b.files) <= set(a.files), (name, sorted(set(a.files) ^ set(b.files))[:5])
        for k in b.files:
            if a[k].dtype.kind in "US" or a[k].ndim == 0:
                if k == "cases":
                    continue                                      # (the TensorFlow branch adds the sculpture pair)
                assert str(a[k]) == str(b[k]), (name, k)
                continue
            if name == "ops.npz":                                 # the custom ops come from oracle/ops_ref.py on both sides
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
            else:                                                 # reference graph code on numpy primitives vs oracle/net_ref.py
                assert a[k].shape == b[k].shape and rel_l1(a[k], b[k]) < 2e-5, (name, k, rel_l1(a[k], b[k]))
    # the bundle its tf.train.Saver stand-in wrote is read back by the product's reader
    from demon_amd import tf_checkpoint as ck
    index, _ = ck.read_index(os.path.join(d, "ckpt", "netRefine_seed1"))
    assert len(index) == 18 and all(n.startswith("netRefine/") for n in index)
