"""metrics of demon_amd.evaluation: known answers (CPU) and the 4-stage prediction protocol (GPU)."""
import numpy as np
import pytest

from demon_amd import evaluation as ev


def test_depth_metrics_known_answers():
    gt = np.array([[1.0, 2.0], [4.0, np.nan]])
    pred = np.array([[2.0, 2.0], [2.0, 3.0]])
    e = ev.depth_errors(pred, gt)
    assert e["pixels"] == 3
    np.testing.assert_allclose(e["l1"], (1 + 0 + 2) / 3)
    np.testing.assert_allclose(e["l1_inverse"], (0.5 + 0 + 0.25) / 3)
    np.testing.assert_allclose(e["abs_relative"], (1 / 1 + 0 + 2 / 4) / 3)
    np.testing.assert_allclose(e["rmse"], np.sqrt((1 + 0 + 4) / 3))
    ld = np.log([2.0, 1.0, 0.5])
    np.testing.assert_allclose(e["scale_invariant"], np.sqrt((ld ** 2).mean() - ld.mean() ** 2))
    np.testing.assert_allclose(e["a1"], 1 / 3)
    # scale invariance: scaling the prediction does not change sc-inv (Eigen et al. eq. 3)
    np.testing.assert_allclose(ev.scale_invariant(3.7 * pred[:1], gt[:1]), ev.scale_invariant(pred[:1], gt[:1]), atol=1e-12)
    for mode in ("abs", "log", "inv"):  # d2 = 2*d1 exactly -> every estimator returns 2
        np.testing.assert_allclose(ev.depth_scale_factor([1.0, 2.0, 5.0], [2.0, 4.0, 10.0], mode), 2.0)
    np.testing.assert_allclose(ev.depth_scale_factor([1.0, 2.0], [2.0, 2.0], "abs"), (2 + 4) / 5.0)


def test_motion_metrics_known_answers():
    z = np.zeros(3)
    m = ev.motion_errors([0, 0, 0.1], [1, 0, 0], [0, 0, 0.3], [0, 2, 0])
    np.testing.assert_allclose(m["rotation_deg"], np.degrees(0.2), rtol=1e-9)
    np.testing.assert_allclose(m["translation_angle_deg"], 90.0)
    np.testing.assert_allclose(m["translation_distance"], np.sqrt(2))
    assert ev.motion_errors(z, [1, 0, 0], z, [1, 0, 0])["rotation_deg"] == 0
    f = np.zeros((2, 3, 3)); g = f.copy(); g[0] = 3; g[1] = 4; g[0, 0, 0] = np.nan
    np.testing.assert_allclose(ev.flow_epe(f, g), 5.0)


def test_metrics_match_reference_goldens():
    """demon_amd.evaluation == the reference's own evaluation/metrics.py, run in the build container on the sculpture depth
    maps the reference ships (tests/golden/make_golden_metrics.py -> metrics_golden.json; inputs regenerated from seeds)"""
    import json
    import os
    import sys
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gdir)
    try:
        import make_golden_metrics as mk
    finally:
        sys.path.remove(gdir)
    with open(os.path.join(gdir, "metrics_golden.json")) as f:
        gold = json.load(f)
    geo = np.load(os.path.join(gdir, "sculpture_geometry.npz"))   # holds the reference's sculpture_depth{1,2}.npy
    names = {"a1": "ratio_threshold_1.25", "a2": "ratio_threshold_1.5625", "a3": "ratio_threshold_1.953125", "pixels": "num_valid"}

    def check(mine, ref):
        for k, v in mine.items():
            want = ref[names.get(k, k)]
            np.testing.assert_allclose(v, np.nan if want is None else want, rtol=2e-4, err_msg=k)   # reference sums in float32

    it = iter(gold["evaluate_depth"])
    for ci, (gt, pred, t) in enumerate(mk.cases(geo["depth1"], geo["depth2"])):
        for scaling, inverse in (("abs", False), ("log", False), ("inv", False), ("abs", True)):
            g = next(it)
            assert g["scaling"] == scaling and g["inverse"] == inverse
            e, es = ev.evaluate_depth(t, gt, pred, inverse_gt=inverse, inverse_pred=inverse, depth_scaling=scaling)
            check(e, g["errs"])
            check(es, g["errs_scaled"])
        m = ev.valid_depth_mask(pred, gt)
        for s_ in ("abs", "log", "inv"):
            np.testing.assert_allclose(ev.depth_scale_factor(pred[m], gt[m], s_), gold["scale"][ci][s_], rtol=2e-4)
    mc = mk.motion_cases()
    it = iter(gold["motion"])
    for i in range(len(mc)):
        for norm in (True, False):
            a, b = mc[i], mc[(i + 1) % len(mc)]
            want = next(it)
            got = ev.motion_errors(a[0:3], a[3:6], b[0:3], b[3:6], normalize_translations=norm)
            np.testing.assert_allclose([got["rotation_deg"], got["translation_distance"], got["translation_angle_deg"]], want,
                                       rtol=1e-6, atol=1e-6)
    fa, fb = mk.flow_cases()
    np.testing.assert_allclose(ev.flow_epe(fa, fb), gold["flow_epe"], rtol=1e-5)


@pytest.mark.gpu
def test_four_stage_protocol(gpu_ctx):
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from conftest import make_inputs
    pair, _ = make_inputs(2, seed=21)
    img2_2 = ev.median_downsample_image2(gpu_ctx, np.ascontiguousarray(pair[:, 3:6]))
    assert img2_2.shape == (2, 3, 48, 64)
    stages = ev.predict_pair(gpu_ctx, pair, img2_2, iterations=3)
    assert len(stages) == 4
    for s in stages:
        assert s["predict_depth0"].shape == (2, 1, 192, 256) and np.isfinite(s["predict_depth0"]).all()
    full = gpu_ctx.full(pair, img2_2, iterations=3)
    np.testing.assert_array_equal(stages[-1]["predict_depth0"], full["predict_depth0"])
