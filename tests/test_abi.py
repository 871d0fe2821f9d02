"""CPU checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol that
include/demon_hip.h declares; the Python variable table matches SURVEY appendix B; host-side API
mirror raises like the reference on bad shapes.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "demon_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(demon_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from demon_amd import build, _lib
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libdemon_hip.so does not export %s" % s
    # the ctypes binding covers exactly the header
    assert sorted(_lib.SIGNATURES) == syms
    _lib.load()


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    from demon_amd import DemonContext, DemonError
    try:
        DemonContext(0, 1)
    except DemonError as e:
        assert "HIP" in str(e) or "device" in str(e)
    else:
        raise AssertionError("context creation must fail without a GPU (no CPU fallback)")


def test_variable_table():
    from demon_amd import weights
    v = weights.variable_shapes()
    assert len(v) == 242
    assert sum(int(np.prod(s)) for s in v.values()) == 45753883
    assert v["netFlow1/conv1y/kernel"] == (9, 1, 6, 32) and v["netFlow1/conv1x/kernel"] == (1, 9, 32, 32)
    assert v["netFlow1/conv2y/kernel"] == (7, 1, 32, 64) and v["netFlow2/conv2y/kernel"] == (7, 1, 32, 32)
    assert v["netFlow1/conv5y/kernel"] == (5, 1, 256, 512) and v["netDM1/conv5y/kernel"] == (3, 1, 256, 512)
    assert v["netFlow1/refine3/upconv/kernel"] == (4, 4, 128, 514) and v["netDM1/refine3/upconv/kernel"] == (4, 4, 128, 512)
    assert v["netFlow1/upsample_flow5to4/upconv/kernel"] == (4, 4, 2, 4)
    assert v["netFlow2/conv2_extra_inputsy/kernel"] == (3, 1, 9, 32)
    assert v["netDM1/conv2_extra_inputsy/kernel"] == (3, 1, 7, 32) and v["netDM2/conv2_extra_inputsy/kernel"] == (3, 1, 8, 32)
    assert v["netDM1/motion_fc1/kernel"] == (6144, 1024) and v["netDM2/motion_fc3/bias"] == (7,)
    assert v["netRefine/predict_depth0/conv2/kernel"] == (3, 3, 16, 1)
    v5 = weights.variable_shapes(480, 640)
    assert v5["netDM1/motion_fc1/kernel"] == (38400, 1024)
    w = weights.synthetic_weights(seed=1)
    w2 = weights.synthetic_weights(seed=1)
    assert all(np.array_equal(w[k], w2[k]) for k in w) and set(w) == set(v)


def test_npz_round_trip(tmp_path):
    from demon_amd import weights
    w = {"a/b/kernel": np.arange(6, dtype=np.float32).reshape(2, 3), "a/b/bias": np.ones(3, np.float32)}
    p = str(tmp_path / "w.npz")
    weights.save_npz(p, w)
    r = weights.load_npz(p)
    assert set(r) == set(w) and all(np.array_equal(r[k], w[k]) for k in w)


def test_drop_in_module_path_exports_the_three_classes():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "python"))
    import importlib
    mod = importlib.import_module("depthmotionnet.networks_original")
    assert sorted(mod.__all__) == ["BootstrapNet", "IterativeNet", "RefinementNet"]
    from depthmotionnet.helpers import angleaxis_to_rotation_matrix
    R = angleaxis_to_rotation_matrix(np.array([-0.03653, 0.26291, 0.06665]))
    g = np.load(os.path.join(ROOT, "tests", "golden", "sculpture_geometry.npz"))
    np.testing.assert_allclose(R, g["Rt2"][:, :3], atol=2e-5)
