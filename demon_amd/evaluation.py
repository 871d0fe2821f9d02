"""Caller-side pieces of the reference's benchmark protocol (examples/evaluation.py) on the MI355X-native path.

* `median_downsample_image2` -- image2_2 as evaluation.py:173 builds it (two 3x3-median downsamples, HIP kernel)
* `predict_pair`             -- the 4-stage prediction loop of evaluation.py:225-256 (bootstrap, 3 x iterative, refinement
                                after every stage) returning what create_prediction_file stores per sample
* depth / motion metrics     -- numpy restatement of the formulas of python/depthmotionnet/evaluation/metrics.py
                                (:40-237 depth errors after Eigen et al., :282-321 scale factor, :390-445 motion errors);
                                the reference module needs minieigen, which is not installable here.
"""
import numpy as np


def median_downsample_image2(ctx, image2):
    """[N,3,192,256] -> [N,3,48,64] (examples/evaluation.py:173)"""
    return ctx.median3x3_downsample(ctx.median3x3_downsample(image2))


def predict_pair(ctx, image_pair, image2_2, iterations=3):
    """Returns a list with one dict per stage (stage 0 = bootstrap, 1..iterations = iterative), each holding the
    network outputs plus 'predict_depth0' from the refinement net run on that stage's depth (evaluation.py:225-256)."""
    image1 = np.ascontiguousarray(image_pair[:, 0:3])
    stages = []
    r = ctx.bootstrap(image_pair, image2_2)
    for it in range(iterations + 1):
        if it > 0:
            r = ctx.iterative(image_pair, image2_2, r["predict_depth2"], r["predict_normal2"], r["predict_rotation"],
                              r["predict_translation"])
        s = dict(r)
        s["predict_depth0"] = ctx.refine(image1, r["predict_depth2"])["predict_depth0"]
        stages.append(s)
    return stages


# ---- depth metrics (metrics.py:24-237) ---------------------------------------------------------------------------
def valid_depth_mask(d1, d2=None):
    m = np.isfinite(d1) & (np.where(np.isfinite(d1), d1, 0) > 0)
    if d2 is not None:
        m &= np.isfinite(d2) & (np.where(np.isfinite(d2), d2, 0) > 0)
    return m


def _pair(d1, d2):
    d1, d2 = np.asarray(d1, np.float64).ravel(), np.asarray(d2, np.float64).ravel()
    assert d1.shape == d2.shape and np.all(np.isfinite(d1) & np.isfinite(d2) & (d1 > 0) & (d2 > 0))
    return d1, d2


def l1(d1, d2):
    d1, d2 = _pair(d1, d2)
    return np.nan if d1.size == 0 else np.abs(d1 - d2).sum() / d1.size


def l1_inverse(d1, d2):
    d1, d2 = _pair(d1, d2)
    return np.nan if d1.size == 0 else np.abs(1.0 / d1 - 1.0 / d2).sum() / d1.size


def rmse(d1, d2):
    d1, d2 = _pair(d1, d2)
    return np.nan if d1.size == 0 else np.sqrt(np.square(d1 - d2).sum() / d1.size)


def rmse_log(d1, d2):
    d1, d2 = _pair(d1, d2)
    return np.nan if d1.size == 0 else np.sqrt(np.square(np.log(d1) - np.log(d2)).sum() / d1.size)


def scale_invariant(d1, d2):
    d1, d2 = _pair(d1, d2)
    if d1.size == 0:
        return np.nan
    ld = np.log(d1) - np.log(d2)
    return np.sqrt(max(np.square(ld).sum() / d1.size - np.square(ld.sum()) / (d1.size ** 2), 0.0))


def abs_relative(pred, gt):
    pred, gt = _pair(pred, gt)
    return np.nan if pred.size == 0 else (np.abs(pred - gt) / gt).sum() / pred.size


def sq_relative(pred, gt):
    pred, gt = _pair(pred, gt)
    return np.nan if pred.size == 0 else (np.square(pred - gt) / gt).sum() / pred.size


def avg_log10(d1, d2):
    d1, d2 = _pair(d1, d2)
    return np.nan if d1.size == 0 else np.abs(np.log10(d1) - np.log10(d2)).sum() / d1.size


def ratio_threshold(d1, d2, threshold):
    d1, d2 = _pair(d1, d2)
    return np.nan if d1.size == 0 else (np.maximum(d1 / d2, d2 / d1) < threshold).sum() / d1.size


def depth_scale_factor(d1, d2, scaling="abs"):
    """scale for d1 that minimises the squared error to d2 (metrics.py:282-321): 'abs' on depth
    (sum d1*d2 / sum d1*d1), 'log' on log depth (exp mean(log d2 - log d1)), 'inv' on inverse depth"""
    d1, d2 = _pair(d1, d2)
    if scaling == "abs":
        den = (d1 * d1).sum()
        return (d1 * d2).sum() / den if den > 0 else 1.0
    if scaling == "log":
        return np.exp((np.log(d2) - np.log(d1)).mean())
    if scaling == "inv":
        den = (1.0 / (d1 * d1)).sum()
        return 1.0 / ((1.0 / (d1 * d2)).sum() / den) if den > 0 else 1.0
    raise ValueError("unknown depth scaling " + scaling)


def depth_errors(pred, gt):
    """dict of all depth distances on the commonly valid pixels"""
    m = valid_depth_mask(pred, gt)
    p, g = np.asarray(pred)[m], np.asarray(gt)[m]
    return {"l1": l1(p, g), "l1_inverse": l1_inverse(p, g), "rmse": rmse(p, g), "rmse_log": rmse_log(p, g),
            "scale_invariant": scale_invariant(p, g), "abs_relative": abs_relative(p, g), "sq_relative": sq_relative(p, g),
            "avg_log10": avg_log10(p, g), "a1": ratio_threshold(p, g, 1.25), "a2": ratio_threshold(p, g, 1.25 ** 2),
            "a3": ratio_threshold(p, g, 1.25 ** 3), "pixels": int(m.sum())}


def evaluate_depth(translation_gt, depth_gt, depth_pred, inverse_gt=True, inverse_pred=True, depth_scaling="abs"):
    """metrics.py:324-374: errors of the prediction on the commonly valid pixels, without and with the least-squares scale.
    Inputs are inverse depths by default; the ground truth is divided by |translation_gt| when that is not 1.
    Returns (errs, errs_pred_scaled) as dicts of depth_errors()."""
    depth_gt, depth_pred = np.asarray(depth_gt), np.asarray(depth_pred)
    m = valid_depth_mask(depth_pred, depth_gt)
    p, g = depth_pred[m].astype(np.float64), depth_gt[m].astype(np.float64)
    if inverse_gt:
        g = 1.0 / g
    if inverse_pred:
        p = 1.0 / p
    tn = float(np.sqrt(np.dot(translation_gt, translation_gt)))
    if not np.isclose(1.0, tn):
        g = g / tn
    errs = depth_errors(p, g)
    scale = depth_scale_factor(*[a[valid_depth_mask(p, g)] for a in (p, g)], scaling=depth_scaling)
    return errs, depth_errors(p * scale, g)


# ---- motion metrics (metrics.py:390-445) -------------------------------------------------------------------------
def _rotmat(aa):
    aa = np.asarray(aa, np.float64).reshape(3)
    angle = np.linalg.norm(aa)
    if angle <= 1e-6:
        return np.eye(3)
    u = aa / angle
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.cos(angle) * np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * np.outer(u, u)


def motion_errors(pred_rotation, pred_translation, gt_rotation, gt_translation, normalize_translations=True):
    """metrics.py:390-445: rotation angular distance (deg), distance of the (optionally normalised) translations, and
    the angle between them (deg; acos of the clipped dot product of the vectors as they are after normalisation)"""
    Rd = _rotmat(pred_rotation).T @ _rotmat(gt_rotation)
    rot_err = np.degrees(np.arccos(np.clip((np.trace(Rd) - 1) / 2, -1, 1)))
    tp, tg = np.asarray(pred_translation, np.float64).reshape(3), np.asarray(gt_translation, np.float64).reshape(3)
    if normalize_translations:
        tg = tg / np.linalg.norm(tg)
        if np.linalg.norm(tp) > 1e-6:
            tp = tp / np.linalg.norm(tp)
    return {"rotation_deg": float(rot_err), "translation_distance": float(np.linalg.norm(tg - tp)),
            "translation_angle_deg": float(np.degrees(np.arccos(np.clip(np.dot(tg, tp), -1, 1))))}


def flow_epe(flow1, flow2):
    """mean end point error (metrics.py:377-387); flows [2,H,W].  The reference masks the error map with its depth validity
    test, i.e. it averages over the pixels whose end point error is finite AND > 0."""
    f1, f2 = np.asarray(flow1, np.float64), np.asarray(flow2, np.float64)
    epe = np.sqrt(np.square(f1 - f2).sum(0))
    m = valid_depth_mask(epe)
    return np.nan if not m.any() else epe[m].mean()
