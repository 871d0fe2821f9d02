"""Generates tests/golden/wiring_original.npz and wiring_v2.npz by RUNNING THE REFERENCE'S OWN GRAPH-BUILDING CODE in this container.

  code   : /root/reference/python/depthmotionnet/{helpers,blocks_original,networks_original}.py and v2/{helpers,blocks,networks}.py,
           imported unmodified; `tensorflow` / `lmbspecialops` resolve to the TensorFlow-1.4 graph emulator oracle/tf1/ (numpy
           restatement of the TensorFlow primitives + oracle/ops_ref.py for the custom ops -- see its header for what that pins:
           the WIRING and the VARIABLE TABLE come from executing reference code, the primitive arithmetic does not)
  inputs : two seeded synthetic pairs (original model: tools/golden_common.synthetic_pair(2, 5), the nets built with batch_size = 2
           so that the reference's batch-dependent reshapes run; v2, whose constructors fix the batch size at 1: synthetic_pair(1, 5)
           and (1, 6)) and, for the original model, the sculpture pair of the reference's fixtures
           (tests/golden/sculpture_inputs.npz, written by make_golden_inputs.py); seeded weights by variable NAME
           (tools/golden_common.seeded_weights on the names / shapes the reference code created)
  outputs: the variable table (name, shape) as the reference code created it, and every fetch of BootstrapNet.eval, 3 x
           IterativeNet.eval, RefinementNet.eval (networks_original.py:60-88, :154-198, :236-255; v2/networks.py) stage by stage --
           for the original model in BOTH data formats (they must agree: the files store channels_first, and the maximum difference
           of the channels_last graph as `nhwc_max_abs_diff`)

Run:  python tests/golden/make_golden_wiring.py      (needs /root/reference; the GPU box only uses the .npz files)
tests/test_wiring_goldens.py holds oracle/net_ref.py (CPU), demon_amd.weights' variable table (CPU) and the HIP path (GPU) to them.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DEMON_REFERENCE", "/root/reference")
SEED, INPUT_SEED, BATCH = 1, 5, 2


def _paths():
    for p in (os.path.join(REF, "python"), os.path.join(ROOT, "tools"), ROOT, os.path.join(ROOT, "oracle", "tf1")):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)


def _nchw(df, a):
    return a if df == "channels_first" or a.ndim != 4 else a.transpose(0, 3, 1, 2)


def _to(df, a):
    return a if df == "channels_first" or a.ndim != 4 else a.transpose(0, 2, 3, 1)


def run(version, df, cases):
    """-> (variables, {case: {stage/key: NCHW array}})"""
    import tensorflow as tf
    import golden_common as G
    assert tf.EMULATED
    with tf.Graph().as_default():
        session = tf.Session()
        results = {}
        variables = None
        for case, (pair, img2_2) in cases:
            n = pair.shape[0]
            # one graph per batch size, like the reference's constructors demand (placeholders have static shapes)
            with tf.Graph().as_default():
                session = tf.Session()
                if version == 1:
                    from depthmotionnet.networks_original import BootstrapNet, IterativeNet, RefinementNet
                    boot, it, ref = BootstrapNet(session, df, n), IterativeNet(session, df, n), RefinementNet(session, df, n)
                else:
                    from depthmotionnet.v2.networks import BootstrapNet, IterativeNet, RefinementNet
                    assert n == 1   # v2/networks.py:21-36: the v2 constructors fix the batch size at 1
                    boot, it, ref = BootstrapNet(session), IterativeNet(session), RefinementNet(session)
                vs = [(v.name.split(":")[0], tuple(v.get_shape().as_list())) for v in tf.global_variables()]
                assert variables is None or variables == vs
                variables = vs
                w = G.seeded_weights(vs, SEED, consistent_flow=(version == 2))
                for v in tf.global_variables():
                    v.load(w[v.name.split(":")[0]], session)
                out = {}
                r = boot.eval(_to(df, pair), _to(df, img2_2))
                out.update(("bootstrap/" + k, _nchw(df, v)) for k, v in r.items())
                for i in range(3):
                    r = it.eval(_to(df, pair), _to(df, img2_2), r["predict_depth2"], r["predict_normal2"], r["predict_rotation"], r["predict_translation"])
                    out.update(("iterative%d/%s" % (i, k), _nchw(df, v)) for k, v in r.items())
                if version == 1:
                    q = ref.eval(_to(df, pair[:, 0:3]), r["predict_depth2"])
                else:
                    q = ref.eval(_to(df, pair[:, 0:3]), r["predict_depth2"], r["predict_normal2"])
                out.update(("refine/" + k, _nchw(df, v)) for k, v in q.items())
                results[case] = out
    return variables, results


def cases_for(version):
    import golden_common as G
    if version == 2:
        return [("synthetic", G.synthetic_pair(1, INPUT_SEED)), ("synthetic_b", G.synthetic_pair(1, INPUT_SEED + 1))]
    cases = [("synthetic", G.synthetic_pair(BATCH, INPUT_SEED))]
    if version == 1:
        g = np.load(os.path.join(HERE, "sculpture_inputs.npz"))
        i1 = (g["image1_u8"].astype(np.float32) / 255 - 0.5).transpose(2, 0, 1)[None]
        i2 = (g["image2_u8"].astype(np.float32) / 255 - 0.5).transpose(2, 0, 1)[None]
        cases.append(("sculpture", (np.concatenate([i1, i2], axis=1), g["image2_2_channels_first_pil"])))
    return cases


def main():
    _paths()
    import tensorflow as tf
    for version, name in ((1, "wiring_original.npz"), (2, "wiring_v2.npz")):
        cases = cases_for(version)
        variables, res = run(version, "channels_first", cases)
        blob = {"variables": np.array(json.dumps([[n, list(s)] for n, s in variables])), "seed": np.array(SEED),
                "input_seed": np.array(INPUT_SEED), "batch": np.array(BATCH), "cases": np.array(json.dumps([c for c, _ in cases])),
                "backend": np.array("reference graph code on oracle/tf1 (tensorflow %s)" % tf.__version__)}
        worst = 0.0
        if version == 1:   # the channels_last graph of the same code (v2/networks.py builds channels_first graphs only)
            variables_l, res_l = run(version, "channels_last", cases)
            assert variables_l == variables
            for case in res:
                for k in res[case]:
                    worst = max(worst, float(np.abs(res[case][k] - res_l[case][k]).max()))
        blob["nhwc_max_abs_diff"] = np.array(worst)
        for case, (pair, img2_2) in cases:
            if not case.startswith("synthetic"):
                blob[case + "/in/image_pair"], blob[case + "/in/image2_2"] = pair, img2_2
            for k, v in res[case].items():
                blob["%s/out/%s" % (case, k)] = np.asarray(v, np.float32)
        np.savez_compressed(os.path.join(HERE, name), **blob)
        print("%s: %d variables, %d floats, cases %s, %d arrays, channels_last max |diff| %.2e, %d bytes" % (
            name, len(variables), sum(int(np.prod(s)) for _, s in variables), [c for c, _ in cases],
            sum(len(r) for r in res.values()), worst, os.path.getsize(os.path.join(HERE, name))))


if __name__ == "__main__":
    main()
