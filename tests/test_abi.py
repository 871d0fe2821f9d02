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


def test_shipped_plans_are_well_formed():
    """demon_amd/tuned/*.json: every entry is (kind, tile, ksplit) of a kernel family that exists, a chained pair (kind 6 / 7 on the
    k x 1 layer) has its 1 x k partner in the plan, and split-K stays on the reduce launch in the shipped plans"""
    import glob
    import json
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "demon_amd")
    src = open(os.path.join(root, "csrc", "internal.h")).read()
    limits = {0: 8, 1: 9, 3: 1, 4: int(re.search(r"STREAM_VARIANTS\s*=\s*(\d+)", src).group(1)), 5: int(re.search(r"FRAG_VARIANTS\s*=\s*(\d+)", src).group(1))}
    limits[6], limits[7] = limits[5], limits[4]
    limits[8] = int(re.search(r"WINO_VARIANTS\s*=\s*(\d+)", src).group(1))      # minimal-filtering transposed conv (conv_wino.hip)
    limits[10] = int(re.search(r"WINO1D_VARIANTS\s*=\s*(\d+)", src).group(1))   # 1-D minimal filtering; 9 is not a kernel
    limits[15] = int(re.search(r"WINO3_VARIANTS\s*=\s*(\d+)", src).group(1))    # 3 x 3 stride 1, transformed input rows stationary (conv_wino3.hip)
    limits[16] = int(re.search(r"WINO4_VARIANTS\s*=\s*(\d+)", src).group(1))    # k x 1 / 1 x k, four outputs per window (conv_wino4.hip)
    limits[13] = 1                                                               # 1 x 7 / 1 x 9 stride-2 conv out of LDS (conv_row.hip)
    limits[12] = 1                                                               # first layer, weights in registers (conv_thin.hip)
    limits[11] = 2                                                               # weight-streaming dense layer (dense_stream.hip)
    files = sorted(glob.glob(os.path.join(root, "tuned", "plan_*.json")))
    assert files
    for f in files:
        d = json.load(open(f))
        assert d["plan"] and d["batch"] >= 1 and d["height"] % 32 == 0 and d["width"] % 32 == 0, f
        for name, (kind, tile, ks) in d["plan"].items():
            assert kind in limits and 0 <= tile < limits[kind] and ks >= 0, (f, name)
            if kind in (0, 4, 5, 8, 10, 11):
                assert 1 <= ks < 1000, (f, name)
            if kind in (6, 7):
                assert name.endswith("y") and name[:-1] + "x" in d["plan"] and ks == 1, (f, name)


def test_kernel_names_round_trip():
    """profile tags <-> rocprofv3 template names (bench.py joins its hip-event table with profiles/*_kernel_stats.csv on them)"""
    from demon_amd.kernel_names import frag_variants, kernel_tag, rocprof_kernel_name
    assert len(frag_variants()) == 22
    cases = {
        "void demon::conv_frag_kernel<4, 1, 1, 1, 1, 1, false>(demon::StreamArgs)": "conv_frag<128x32,v6>",
        "void demon::conv_frag_chain_kernel<1, 4, 2, 1, 1>(demon::StreamArgs, demon::StreamArgs)": "conv_frag_chain<64x128,v5>",
        "void demon::conv_stream_chain_kernel<4, 2, 1>(demon::StreamArgs, demon::StreamArgs)": "conv_stream_chain<256x32,w4>",
        "void demon::conv_stream_kernel<4, 2, 1, 1, false>(demon::StreamArgs)": "conv_stream<256x32,w4k1>",
        "void demon::conv_patch_kernel<64, 2, 2, 1, 1, 4, 4, 2>(demon::PatchArgs)": "conv_patch<64x64,t4>",
        "void demon::conv_mfma_kernel<128, 32, 4, 1, false>(demon::ConvArgs)": "conv_mfma<128x32>",
        "void demon::deconv4_kernel<32, 1, 4, 4>(demon::PatchArgs)": "deconv4<32x128>",
        "void demon::wino_deconv_kernel<2, 4, 3, 1, 1>(demon::WinoArgs)": "wino_deconv<16x32>",
        "void demon::wino_deconv_kernel<1, 2, 3, 2, 1>(demon::WinoArgs)": "wino_deconv<32x16>",
        "void demon::wino_deconv_kernel<3, 10, 1, 1, 2>(demon::WinoArgs)": "wino_deconv<16x48,kh2>",
        "void demon::wino1d_kernel<1, 0, 2, 2, 2, 1, false>(demon::Wino1Args)": "wino1d<t5,v0>",
        "void demon::wino1d_kernel<0, 1, 4, 1, 4, 2, false>(demon::Wino1Args)": "wino1d<t3,v5>",
        "void demon::wino1d_kernel<0, 0, 4, 1, 3, 4, false>(demon::Wino1Args)": "wino1d<t3,v8>",
        "void demon::wino1d_kernel<1, 0, 2, 2, 3, 2, true>(demon::Wino1Args)": "wino1d<t5,v9>",
        "void demon::dense_stream_kernel<true, 0>(demon::DenseArgs)": "dense_stream<128x32,v1>",
        "void demon::conv_thin_kernel<9, 2, 14>(demon::ThinArgs)": "conv_thin<32x512,t9>",
        "void demon::conv_row_kernel<3, true>(demon::RowArgs)": "conv_row<32x128,t9>",
        "void demon::conv_row_kernel<2, false>(demon::RowArgs)": "conv_row<32x128,t7>",
    }
    for name, tag in cases.items():
        assert kernel_tag(name) == tag
        shown = rocprof_kernel_name(tag + "+splitk")
        assert shown.split("<")[0] == "demon::" + tag.split("<")[0] + "_kernel"
        assert shown.split("<")[0].replace("demon::", "") in name
    # the minimal-filtering kernels whose tag family is not the kernel's name
    more = {
        "void demon::wino3_rows_kernel<2, 2, 4, 1, false, 0>(demon::Wino3Args)": "wino3rows<t3x3,v0>",
        "void demon::wino3_rows_kernel<4, 1, 3, 1, false, 1>(demon::Wino3Args)": "wino3rows<f4t3x3,v11>",
        "void demon::wino3_rows_kernel<8, 1, 2, 1, false, 2>(demon::Wino3Args)": "wino3rows<s2t3x3,v19>",
        "void demon::wino4_kernel<1, 0, 4, 2, 2, 2, false, false, false>(demon::Wino4Args)": "wino4<t5,v4>",
        "void demon::wino4_kernel<0, 1, 2, 2, 2, 1, false, false, false>(demon::Wino4Args)": "wino4<t3,v8>",
        "void demon::wino4_kernel<0, 0, 4, 2, 3, 2, false, false, false>(demon::Wino4Args)": "wino4<t3,v9>",
        "void demon::wino4_kernel<0, 1, 4, 1, 2, 2, false, false, true>(demon::Wino4Args)": "wino4<t3,v1,walk>",
    }
    for name, tag in more.items():
        assert kernel_tag(name) == tag


def test_pmc_family_covers_every_contraction_kernel():
    """tools/pmc_summary.py averages the HBM counters over the launches bench.py's `roofline_family` counts; its list of kernel names must
    know every family that bench.py names there (round 4 shipped a summary that had left the wino3_rows / wino4 launches out)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    src = open(os.path.join(ROOT, "bench.py")).read()
    m = re.search(r'"kernel": "all conv / deconv / dense launches \(([^)]*) kernels\)"', src)
    assert m, "bench.py no longer names the families of roofline_family"
    families = [f.strip() for f in m.group(1).split(",")]
    assert len(families) >= 16
    symbol = {"wino3rows": "wino3_rows_kernel", "deconv4": "deconv4_kernel"}
    for fam in families:
        assert symbol.get(fam, fam + "_kernel") in mod.CONTRACTION_KERNELS, fam
    # and every kernel source that launches a contraction kernel is behind one of the names
    for k in mod.CONTRACTION_KERNELS:
        assert any(k + "<" in open(os.path.join(ROOT, "demon_amd", "csrc", f)).read() or k + "(" in open(os.path.join(ROOT, "demon_amd", "csrc", f)).read()
                   for f in os.listdir(os.path.join(ROOT, "demon_amd", "csrc")) if f.endswith(".hip")), k



def test_hardware_queue_request_is_made_by_the_lanes_module_only():
    """`import demon_amd` leaves the process environment alone; `import demon_amd.lanes` exports GPU_MAX_HW_QUEUES = 8 (16
    under a torch.distributed launcher) unless the caller chose a value or opted out, records what it did, and warns when a context
    already exists (the HIP runtime reads the variable once, at its first call); the C ABI offers the same numbers as a hint"""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = ("import os; import demon_amd; a = os.environ.get('GPU_MAX_HW_QUEUES'); import demon_amd.lanes as L; "
            "print(a, os.environ.get('GPU_MAX_HW_QUEUES'), L.HW_QUEUES['set_by'], L.HW_QUEUES['runtime_was_up'])")
    for env_extra, want in (({}, "None 8 demon_amd.lanes False"), ({"GPU_MAX_HW_QUEUES": "4"}, "4 4 caller False"), ({"DEMON_HW_QUEUES": "0"}, "None None None False"),
                            ({"DEMON_HW_QUEUES": "6"}, "None 6 demon_amd.lanes False"), ({"LOCAL_RANK": "0"}, "None 16 demon_amd.lanes False"),
                            ({"LOCAL_RANK": "0", "DEMON_HW_QUEUES": "8"}, "None 8 demon_amd.lanes False")):
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "DEMON_HW_QUEUES", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.strip() == want, (env_extra, r.stdout, r.stderr[-500:])
    # too late: a context was created before the module was imported -> the request is recorded as ineffective, with a warning
    late = ("import warnings; from demon_amd import DemonContext; DemonContext.created_in_process = 1\n"
            "with warnings.catch_warnings(record=True) as w:\n    warnings.simplefilter('always'); import demon_amd.lanes as L\n"
            "print(L.HW_QUEUES['runtime_was_up'], len(w), 'no effect' in str(w[0].message))")
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "DEMON_HW_QUEUES", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, "-c", late.replace("\\n", "\n")], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "True 1 True", (r.stdout, r.stderr[-800:])
    from demon_amd import _lib
    lib = _lib.load()
    assert lib.demon_hw_queues_hint(0) == 8 and lib.demon_hw_queues_hint(1) == 16
