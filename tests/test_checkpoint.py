"""TensorBundle reader / writer round trip and the tensorflow stub that lets the reference's example.py run
without TensorFlow (CPU parts)."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_tensorbundle_round_trip(tmp_path):
    from demon_amd import tf_checkpoint as ck
    rng = np.random.default_rng(0)
    tensors = {"netFlow1/conv1y/kernel": rng.standard_normal((9, 1, 6, 32)).astype(np.float32),
               "netFlow1/conv1y/bias": rng.standard_normal((32,)).astype(np.float32),
               "netDM1/motion_fc3/kernel": rng.standard_normal((128, 7)).astype(np.float32),
               "scalar": np.float32(3.5).reshape(())}
    for i in range(300):  # enough entries for several index blocks
        tensors["net/layer%03d/kernel" % i] = rng.standard_normal((3, 1, 2, 4)).astype(np.float32)
    prefix = str(tmp_path / "ckpt" / "model")
    ck.save_tf_checkpoint(prefix, tensors, block_size=512)
    # structure: footer magic, sorted keys, header entry
    data = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", data[-8:])[0] == 0xDB4775248B80FB57
    index, shards = ck.read_index(prefix)
    assert shards == 1 and set(index) == set(tensors)
    assert index["netFlow1/conv1y/kernel"]["shape"] == [9, 1, 6, 32] and index["netFlow1/conv1y/kernel"]["dtype"] == 1
    got = ck.load_tf_checkpoint(prefix)
    assert set(got) == set(tensors)
    for k in tensors:
        assert got[k].shape == tensors[k].shape and np.array_equal(got[k], tensors[k]), k
    sub = ck.load_tf_checkpoint(prefix, ["netFlow1/conv1y/bias"])
    assert list(sub) == ["netFlow1/conv1y/bias"]
    with pytest.raises(KeyError):
        ck.load_tf_checkpoint(prefix, ["missing/variable"])
    # crc32c known answer (RFC 3720): "123456789" -> 0xE3069283
    assert ck._crc32c(b"123456789") == 0xE3069283
    with open(prefix + ".index", "r+b") as f:  # corrupt the magic
        f.seek(-1, 2)
        f.write(b"\x00")
    with pytest.raises(ValueError):
        ck.read_index(prefix)


def test_prefix_compressed_blocks_are_decoded():
    """TensorFlow's table builder shares key prefixes between restart points; build such a block by hand"""
    from demon_amd import tf_checkpoint as ck
    keys = [b"netFlow1/conv1x/bias", b"netFlow1/conv1x/kernel", b"netFlow1/conv1y/bias"]
    body, prev = bytearray(), b""
    for i, k in enumerate(keys):
        shared = 0
        while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
            shared += 1
        if i == 0:
            shared = 0
        v = b"v%d" % i
        body += ck._write_varint(shared) + ck._write_varint(len(k) - shared) + ck._write_varint(len(v)) + k[shared:] + v
        prev = k
    body += struct.pack("<II", 0, 1)  # one restart at 0
    data = bytes(body) + b"\x00" + b"\x00\x00\x00\x00"
    assert ck._read_block(data, 0, len(body)) == [(keys[0], b"v0"), (keys[1], b"v1"), (keys[2], b"v2")]


def test_tensorflow_stub_restores_weights(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "python", "tf_stub"))
    try:
        import importlib
        tf = importlib.import_module("tensorflow")
        assert "stub" in tf.__version__
        from demon_amd import weights as W, tf_checkpoint as ck
        import demon_amd
        from demon_amd import runtime
        runtime._default_weights[1] = runtime._default_weights[2] = None   # the first restored set becomes the default
        w = {k: np.full(s, 0.25, np.float32) for k, s in W.variable_shapes().items()}
        prefix = str(tmp_path / "demon_original")
        ck.save_tf_checkpoint(prefix, w)
        gpu_options = tf.GPUOptions()
        gpu_options.per_process_gpu_memory_fraction = 0.8
        session = tf.InteractiveSession(config=tf.ConfigProto(allow_soft_placement=True, gpu_options=gpu_options))
        session.run(tf.global_variables_initializer())
        tf.train.Saver().restore(session, prefix)
        assert set(session.demon_weights) == set(w)
        assert demon_amd.default_weights() is session.demon_weights
        assert session.demon_weights["netDM2/motion_fc1/kernel"].shape == (6144, 1024)
        with pytest.raises(IOError):
            tf.train.Saver().restore(session, str(tmp_path / "nope"))
        # a checkpoint of the v2 model (example_v2.py:88-89) is recognised by its dense5 layer
        w2 = {k: np.full(s, 0.5, np.float32) for k, s in W.variable_shapes(version=2).items()}
        prefix2 = str(tmp_path / "demon_v2")
        ck.save_tf_checkpoint(prefix2, w2)
        session2 = tf.InteractiveSession()
        tf.train.Saver().restore(session2, prefix2)
        assert set(session2.demon_weights) == set(w2)
        assert session2.demon_weights["netFlow2/dense5/kernel"].shape == (4608, 4608)
        assert demon_amd.default_weights(2) is session2.demon_weights and demon_amd.default_weights(1) is session.demon_weights
    finally:
        sys.path.remove(os.path.join(ROOT, "python", "tf_stub"))
        sys.modules.pop("tensorflow", None)
        from demon_amd import runtime
        runtime._default_weights[1] = runtime._default_weights[2] = None


def test_crc32c_and_entry_checksums(tmp_path):
    """BundleEntryProto field 6 (masked crc32c of the tensor bytes, what TensorFlow's BundleReader verifies): known answers of
    CRC-32C, the vectorised implementation against the byte recurrence, and corruption detection on load"""
    from demon_amd import tf_checkpoint as ck
    assert ck.crc32c(b"123456789") == 0xE3069283           # the standard CRC-32C check value
    assert ck.crc32c(b"") == 0 and ck.crc32c(bytes(32)) == 0x8A9136AA   # RFC 3720 B.4: 32 zero bytes
    rng = np.random.default_rng(5)
    for n in (1, 4095, 4096, 4097, 3 * 4096, 70001):
        d = rng.integers(0, 256, n, dtype=np.uint8)
        assert ck.crc32c(d.tobytes()) == ck._crc_register(d, 0xFFFFFFFF) ^ 0xFFFFFFFF
    w = {"a/kernel": rng.standard_normal((3, 3, 4, 8)).astype(np.float32), "a/bias": rng.standard_normal(8).astype(np.float32)}
    prefix = str(tmp_path / "ck")
    ck.save_tf_checkpoint(prefix, w)
    index, _ = ck.read_index(prefix)
    assert all("crc32c" in index[k] for k in w)
    assert index["a/bias"]["crc32c"] == ck._mask_crc(ck.crc32c(w["a/bias"].tobytes()))
    r = ck.load_tf_checkpoint(prefix, list(w))
    assert all(np.array_equal(r[k], w[k]) for k in w)
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[5] ^= 0x40
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        ck.load_tf_checkpoint(prefix, list(w))
    assert ck.load_tf_checkpoint(prefix, list(w), verify_crc=False)["a/kernel"].shape == (3, 3, 4, 8)
