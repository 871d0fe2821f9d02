"""Parity at BASELINE.json's FULL sizes -- configs[2] (batch 32 @256x192, the metric's configuration, with the shipped launch
plan) and configs[4] (batch 64 @640x480) -- where the CPU oracle cannot run the whole batch in test time.  Checked through
size-independent properties of the path, plus the oracle on a sample of the batch:

  * pairs are independent (no batch statistics anywhere): row i of the big batch == the same pair evaluated alone /
    in a small batch on another context (other launch plans -> summation order only);
  * permutation equivariance: permuting the pairs permutes the outputs, bit for bit;
  * determinism: two runs are bit-identical;
  * the oracle agrees on EVERY pair of the metric's batch (all 32 at configs[2], pair by pair) and on 8 of 64 at configs[4]
    (relative L1 <= 1e-3, BASELINE.json's tolerance);
  * every lane of a calibrated three-lane group (the headline's protocol, throughput-mode plan) agrees with the oracle on every
    pair of its batch;
  * depth -> flow -> depth round trip through the two geometry kernels at full batch and resolution.
"""
import numpy as np
import pytest

from conftest import rel_l1, make_inputs
from oracle import net_ref

pytestmark = pytest.mark.gpu
KEYS = ("predict_flow5", "predict_flow2", "predict_depth2", "predict_normal2", "predict_rotation", "predict_translation",
        "predict_depth0")


def _oracle_every_pair(ref, pair, img2_2, got, indices, chunk=8, iterations=3, tol=1e-3):
    """the CPU oracle on pairs `indices` of a batch, in chunks; every key of every pair within `tol` relative L1"""
    indices = list(indices)
    worst = 0.0
    for at in range(0, len(indices), chunk):
        sel = indices[at:at + chunk]
        want = ref.full(pair[sel], img2_2[sel], iterations=iterations)
        for k in KEYS:
            for j, i in enumerate(sel):
                err = rel_l1(got[k][i], want[k][j])
                assert err < tol, "%s pair %d: rel L1 %.3e" % (k, i, err)
                worst = max(worst, err)
    return worst


def test_config2_batch32_full_pipeline(gpu_ctx, synth_weights):
    from demon_amd import DemonContext
    n = 32
    ctx = DemonContext(0, n, 192, 256)
    try:
        ctx.set_weights(synth_weights)
        assert ctx.load_tuned_plan(n), "the shipped plan for the metric's configuration must load"
        pair, img2_2 = make_inputs(n, seed=40)
        got = ctx.full(pair, img2_2, iterations=3)
        assert all(np.isfinite(got[k]).all() for k in KEYS)
        # determinism
        again = ctx.full(pair, img2_2, iterations=3)
        for k in KEYS:
            np.testing.assert_array_equal(got[k], again[k])
        # permutation equivariance (bit exact: same kernels, same per-pair arithmetic)
        perm = np.random.default_rng(41).permutation(n)
        pgot = ctx.full(pair[perm], img2_2[perm], iterations=3)
        for k in KEYS:
            np.testing.assert_array_equal(pgot[k], got[k][perm])
        # independence: pairs 0..3 of the batch == the same pairs on the small-batch context (heuristic plans)
        small = gpu_ctx.full(pair[:4], img2_2[:4], iterations=3)
        for k in KEYS:
            assert rel_l1(got[k][:4], small[k]) < 1e-4, k
        # the oracle on ALL 32 pairs of the batch, pair by pair (chunks of eight: the CPU restatement stays within seconds)
        _oracle_every_pair(net_ref.DemonRef(synth_weights), pair, img2_2, got, range(n))
    finally:
        ctx.close()


def test_config4_batch64_640x480(synth_weights):
    from demon_amd import DemonContext, weights
    H, W, n = 480, 640, 64
    w = weights.synthetic_weights(seed=1, height=H, width=W)
    big = DemonContext(0, n, H, W)
    one = DemonContext(0, 1, H, W)
    try:
        big.set_weights(w)
        one.set_weights(w)
        big.load_tuned_plan(n)
        pair, img2_2 = make_inputs(n, H, W, seed=42)
        got = big.full(pair, img2_2, iterations=3)
        assert all(np.isfinite(got[k]).all() for k in KEYS)
        again = big.full(pair, img2_2, iterations=3)
        for k in KEYS:
            np.testing.assert_array_equal(got[k], again[k])
        # independence: pair 17 alone on a batch-1 context
        single = one.full(pair[17:18], img2_2[17:18], iterations=3)
        for k in KEYS:
            assert rel_l1(got[k][17:18], single[k]) < 1e-4, k
        # the oracle on eight pairs spread over the batch (first / last / both halves of every 16), two chunks of four
        _oracle_every_pair(net_ref.DemonRef(w), pair, img2_2, got, [0, 9, 17, 26, 34, 45, 55, 63], chunk=4)
        # geometry kernels at the full level-2 size of this config: inverse depth -> flow -> inverse depth
        rng = np.random.default_rng(43)
        h2, w2 = H // 4, W // 4
        inv_depth = (0.2 + rng.random((n, 1, h2, w2))).astype(np.float32)
        rot = (0.05 * rng.standard_normal((n, 3))).astype(np.float32)
        tr = rng.standard_normal((n, 3)).astype(np.float32)
        tr /= np.linalg.norm(tr, axis=1, keepdims=True)
        K = np.array([0.89115971, 1.18821287, 0.5, 0.5], np.float32)
        flow = big.depth_to_flow(inv_depth, K, rot, tr, True, True, False)
        for method in (0, 1):
            back = big.flow_to_depth(flow, K, rot, tr, True, True, method)
            # pixels near the epipole are ill-conditioned in any triangulation: judge in aggregate and by the median
            m = np.isfinite(back) & np.isfinite(flow[:, :1])
            assert m.mean() > 0.99
            assert np.median(np.abs(back[m] - inv_depth[m]) / inv_depth[m]) < 1e-4
    finally:
        big.close()
        one.close()


def test_config2_every_lane_of_a_calibrated_group(synth_weights):
    """The headline's protocol (bench.py): a three-lane LaneGroup at batch 32 on the throughput-mode plan, stream mapping calibrated,
    steps fed round robin.  Every lane holds a different batch; after several rounds with all lanes in flight EVERY lane's outputs
    are held to the CPU oracle on EVERY pair (3 x 32 pairs), and a second round of steps does not change a bit."""
    from demon_amd.lanes import LaneGroup
    n, lanes = 32, 3
    group = LaneGroup(synth_weights, lanes=lanes, batch=n)
    try:
        assert group.ctxs[0].load_tuned_plan(n, lanes=lanes) == n, "the shipped throughput-mode plan of the metric's configuration"
        batches = [make_inputs(n, seed=60 + i) for i in range(lanes)]
        group.upload_inputs(batches)
        rates = group.calibrate(n, iterations=3, steps_per_lane=2)
        assert rates and group.mapping["lanes"] == len(group) >= 1
        group.run_resident(n, 3 * len(group), iterations=3)     # every lane three times, all in flight
        group.synchronize()
        first = [c.download_outputs(n) for c in group.ctxs]
        group.run_resident(n, 2 * len(group), iterations=3)
        group.synchronize()
        ref = net_ref.DemonRef(synth_weights)
        for lane, (c, out, (pair, img2_2)) in enumerate(zip(group.ctxs, first, batches)):
            assert all(np.isfinite(out[k]).all() for k in KEYS), lane
            again = c.download_outputs(n)
            for k in KEYS:
                np.testing.assert_array_equal(again[k], out[k], err_msg="lane %d %s" % (lane, k))
            _oracle_every_pair(ref, pair, img2_2, out, range(n))
    finally:
        group.close()


def test_config2_headline_protocol_every_pair_of_every_kept_lane():
    """bench.py's own set-up, not a stand-in for it (tools/lane_parity.py, in a process of its own like the bench): lane 0 + the
    throughput-mode plan the default `--lanes` resolves to (five requested -> `_l4`), LaneGroup(5).calibrate() with its defaults under
    the package's hardware-queue request, then EVERY pair of EVERY lane the calibration kept against the CPU oracle."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "lane_parity.py"), "--lanes", "5"], cwd=root, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    m = rec["mapping"]
    assert rec["lanes_kept"] == m["lanes"] and 1 <= rec["lanes_kept"] <= 5
    assert m["verified_pairs_per_s"] > 0 and m["attempts"] >= 1 and "reproduced" in m           # the winner was measured again where it was left
    assert rec["hw_queues"]["env"] is not None and rec["hw_queues"]["runtime_was_up"] is False  # the queue request was made in time
    assert rec["outputs_finite"] and rec["second_round_bit_equal"]
    assert rec["pairs_checked"] == 32 * rec["lanes_kept"]
    assert rec["worst_rel_l1"] < 1e-3, rec["worst_at"]
