"""Poison harness: every kernel family x variant on tensors that sit FLUSH between NaN-poisoned neighbours.

libdemon_hip.so, with DEMON_POISON_GUARD=1 in the environment, places every device allocation of a context / of a layer-level call
-- input, output, packed weights, every derived weight form (fragment order, U of the minimal-filtering kernels, re-blocked dense
weights), workspaces -- between two 4 MiB zones of a quiet-NaN canary and drops the slack planes of the activation buffers
(include/demon_hip.h: demon_debug_check_guards).  Then

  * a kernel that reads outside a tensor and USES the value produces NaN / differs from the unguarded run -- the tests demand finite
    results that equal PyTorch (layer level) or the unguarded context bit for bit (whole nets);
  * a kernel that writes outside a tensor changes a canary -- the layer-level entry points fail with DEMON_ERR_HIP by themselves,
    the whole-net tests call check_guards().

Since round 5 the raw buffer resources of the minimal-filtering / first-layer / dense kernels carry the true extent of their tensors
(internal.h: rsrc_bytes), so for them a read past the end returns 0 by hardware; this harness is what covers the other kernels and
what one range check cannot see.

Layer level: EVERY test of tests/test_variants_gpu.py (all families, all variants, ragged shapes) is re-run here under the switch.
"""
import os

import numpy as np
import pytest

import test_variants_gpu as V
from conftest import make_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def poison_guard():
    os.environ["DEMON_POISON_GUARD"] = "1"
    yield
    os.environ.pop("DEMON_POISON_GUARD", None)


# ---- layer level: the whole variants suite again, guarded -------------------------------------------------------------------------
for _name in dir(V):
    if _name.startswith("test_") and callable(getattr(V, _name)):
        globals()["test_poisoned_" + _name[5:]] = getattr(V, _name)
del _name


def test_the_switch_is_live():
    """a context created under the switch reports zero violations after a clean pass and really sits between canaries (the C side
    refuses the check on a context created without it: the harness cannot silently be a no-op)"""
    from demon_amd import DemonContext
    from demon_amd.engine import DemonError
    os.environ.pop("DEMON_POISON_GUARD", None)
    plain = DemonContext(0, 1, 192, 256)
    try:
        with pytest.raises(DemonError, match="DEMON_POISON_GUARD"):
            plain.check_guards()
    finally:
        plain.close()
    os.environ["DEMON_POISON_GUARD"] = "1"
    ctx = DemonContext(0, 1, 192, 256)
    try:
        assert ctx.check_guards()[0] == 0
    finally:
        ctx.close()


# ---- whole nets: every shipped launch plan, guarded context == plain context bit for bit, canaries intact ---------------------------
@pytest.mark.parametrize("n,lanes,version", [(1, 1, 1), (3, 1, 1), (8, 3, 1), (32, 1, 1), (32, 3, 1), (2, 1, 2), (32, 3, 2)])
def test_whole_net_between_poisoned_neighbours(synth_weights, n, lanes, version):
    from demon_amd import DemonContext, weights
    w = synth_weights if version == 1 else weights.synthetic_weights(seed=1, version=2)
    pair, img2_2 = make_inputs(n, seed=90 + n)
    outs = []
    for guarded in (False, True):
        if guarded:
            os.environ["DEMON_POISON_GUARD"] = "1"
        else:
            os.environ.pop("DEMON_POISON_GUARD", None)
        ctx = DemonContext(0, n, 192, 256, version=version)
        try:
            ctx.set_weights(w)
            ctx.load_tuned_plan(n, lanes=lanes)
            if lanes > 1:
                ctx.set_option("side_branches", 0)
            out = ctx.full(pair, img2_2, iterations=3)
            if guarded:
                bad, where = ctx.check_guards()
                assert bad == 0, where
        finally:
            ctx.close()
        outs.append(out)
    plain, poisoned = outs
    for k in plain:
        assert np.isfinite(poisoned[k]).all(), k
        np.testing.assert_array_equal(poisoned[k], plain[k], err_msg=k)
