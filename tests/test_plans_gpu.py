"""Every launch plan shipped in demon_amd/tuned/ against the CPU oracle, plus BASELINE.json configs[1] (batch 8, bootstrap net only).

A plan selects the kernel family / tile / split-K of every layer, so an untested plan is an untested kernel mix.  For each plan
file: the context of exactly that shape / batch / model version loads it (no nearest-batch substitution), runs the whole pipeline
(networks_original.py:22-255 resp. v2/networks.py), and
  * the oracle agrees on >= 4 sampled pairs of the batch (relative L1 <= 1e-3, BASELINE.json north_star),
  * two runs are bit-identical,
  * demon_profile_full reports, layer by layer, the kernel family (and variant, where the tag carries it) the plan names --
    i.e. the plan was really installed and no layer fell back to the heuristics.
"""
import glob
import json
import os
import re

import numpy as np
import pytest

from conftest import rel_l1, make_inputs
from oracle import net_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANS = sorted(glob.glob(os.path.join(ROOT, "demon_amd", "tuned", "plan_*.json")))
KEYS = ("predict_flow5", "predict_conf5", "predict_flow2", "predict_conf2", "predict_depth2", "predict_normal2", "predict_rotation",
        "predict_translation", "predict_scale", "predict_depth0")
FAMILY = {0: ("conv_mfma<",), 1: ("conv_patch<", "deconv4<"), 3: ("conv_small",), 4: ("conv_stream<",), 5: ("conv_frag<",),
          6: ("conv_frag_chain<",), 7: ("conv_stream_chain<",), 8: ("wino_deconv<",), 10: ("wino1d<",), 11: ("dense_stream<",), 12: ("conv_thin<",), 13: ("conv_row<",), 15: ("wino3rows<",), 16: ("wino4<",)}


def _weights(version, height, width):
    from demon_amd import weights
    return weights.synthetic_weights(seed=1, height=height, width=width, version=version)


def check_plan_ran(plan, records):
    """plan: {layer: [kind, tile, ksplit]}, records: demon_profile_full of a pass at the plan's batch size"""
    by_name = {}
    for r in records:
        by_name.setdefault(r["name"], r["kernel"])
    seen = 0
    for layer, (kind, tile, ksplit) in plan.items():
        if layer in by_name:
            tag = by_name[layer]
        else:
            # k x 1 / 1 x k pairs run as one step "<...>y+x": the fused conv1 pair (conv_pair.hip, no plan entry applies) or a
            # chained launch selected by kind 6 / 7 on the k x 1 layer
            pair = layer[:-1] + "y+x" if layer[-1] in "xy" else None
            if pair not in by_name:
                continue   # layers without a step of their own (motion_fc2 / motion_fc3 run inside motion_tail)
            tag = by_name[pair]
            if tag == "conv_pair":
                continue
            assert layer.endswith("y") or kind not in (6, 7), layer
            if layer.endswith("x"):
                continue   # the chain is the k x 1 layer's choice
        assert tag.startswith(FAMILY[kind]), "%s: plan kind %d, ran %s" % (layer, kind, tag)
        if kind in (5, 6):
            assert re.search(r",v%d>" % tile, tag), "%s: plan variant %d, ran %s" % (layer, tile, tag)
        if kind in (0, 4, 5) and ksplit <= 1:
            assert "+" not in tag, "%s: no split-K planned, ran %s" % (layer, tag)
        seen += 1
    assert seen >= 80, "only %d plan entries could be matched to launches" % seen


@pytest.mark.parametrize("path", PLANS, ids=[os.path.basename(p)[5:-5] for p in PLANS])
def test_shipped_plan_matches_oracle(path):
    from demon_amd import DemonContext
    with open(path) as f:
        meta = json.load(f)
    n, H, W, version = meta["batch"], meta["height"], meta["width"], meta.get("model_version", 1)
    w = _weights(version, H, W)
    ctx = DemonContext(0, n, H, W, version=version)
    try:
        ctx.set_weights(w)
        # plans tuned in throughput mode (tools/tune.py --lanes L -> ..._l<L>.json) are what a lane of a group loads
        assert ctx.load_tuned_plan(n, nearest=False, lanes=meta.get("tune_lanes", 1)) == n
        assert ctx.get_plan(n) == {k: list(v) for k, v in meta["plan"].items()}
        pair, img2_2 = make_inputs(n, H, W, seed=100 + n)
        got = ctx.full(pair, img2_2, iterations=3)
        keys = KEYS + (("predict_normal0",) if version == 2 else ())
        assert all(np.isfinite(got[k]).all() for k in keys)
        again = ctx.full(pair, img2_2, iterations=3)
        for k in keys:
            np.testing.assert_array_equal(got[k], again[k])
        # the oracle on (up to) four pairs spread over the batch, first and last included
        sel = sorted(set(np.linspace(0, n - 1, 4).round().astype(int).tolist()))
        ref = net_ref.DemonRef(w) if version == 1 else net_ref.DemonRefV2(w)
        want = ref.full(pair[sel], img2_2[sel], iterations=3)
        for k in keys:
            err = rel_l1(got[k][sel], want[k])
            assert err < 1e-3, "%s rel L1 %.3e" % (k, err)
        check_plan_ran(meta["plan"], ctx.profile_full(n, 3, 1))
    finally:
        ctx.close()


def test_config1_batch8_bootstrap_only():
    """BASELINE.json configs[1]: batch 8, bootstrap net only (networks_original.py:22-88), shipped plan, oracle on ALL pairs"""
    from demon_amd import DemonContext
    n = 8
    w = _weights(1, 192, 256)
    ctx = DemonContext(0, n, 192, 256)
    try:
        ctx.set_weights(w)
        assert ctx.load_tuned_plan(n, nearest=False) == n
        pair, img2_2 = make_inputs(n, seed=81)
        got = ctx.bootstrap(pair, img2_2)
        keys = KEYS[:-1]
        again = ctx.bootstrap(pair, img2_2)
        for k in keys:
            assert np.isfinite(got[k]).all(), k
            np.testing.assert_array_equal(got[k], again[k])
        # device-resident form (what bench.py --workload bootstrap times): same bits
        ctx.upload_inputs(pair, img2_2)
        ctx.run_bootstrap(n)
        ctx.synchronize()
        res = ctx.download_outputs(n, with_depth0=False)
        for k in keys:
            np.testing.assert_array_equal(got[k], res[k])
        want = net_ref.DemonRef(w).bootstrap(pair, img2_2)
        for k in keys:
            err = rel_l1(got[k], want[k])
            assert err < 1e-3, "%s rel L1 %.3e" % (k, err)
            for i in range(n):   # and pair by pair: an error confined to one sample must not hide in the batch sum
                assert rel_l1(got[k][i], want[k][i]) < 1e-3, (k, i)
    finally:
        ctx.close()


def test_ragged_batch_keeps_the_tuned_plan():
    """a last batch smaller than the context's (3 pairs in a context built and tuned for 8, as runtime.get_context sets one up)
    runs on the plan entries of the nearest tuned batch size -- the minimal-filtering / streaming kernels of the batch-8 plan --
    not on the untuned heuristics, and agrees with the oracle"""
    from demon_amd import DemonContext
    w = _weights(1, 192, 256)
    ctx = DemonContext(0, 8, 192, 256)
    try:
        ctx.set_weights(w)
        assert ctx.load_tuned_plan(8, nearest=False) == 8
        pair, img2_2 = make_inputs(3, seed=83)
        got = ctx.full(pair, img2_2, iterations=3)
        tags = [r["kernel"] for r in ctx.profile_full(3, 3, 1)]
        assert any(t.startswith("wino_deconv<") for t in tags) and any(t.startswith("wino1d<") for t in tags), sorted(set(tags))
        want = net_ref.DemonRef(w).full(pair, img2_2, iterations=3)
        for k in KEYS:
            assert rel_l1(got[k], want[k]) < 1e-3, k
    finally:
        ctx.close()
