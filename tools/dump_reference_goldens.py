#!/usr/bin/env python
"""Dumps golden vectors from the REAL reference stack -- TensorFlow 1.4 + lmbspecialops + lmb-freiburg/demon's own
python/depthmotionnet -- so that this repo's tests can pin what they could not pin without it (DESIGN.md section 4 "still
unpinned"): lmbspecialops' warp2d border / non-finite rules, flow_to_depth on inconsistent flow, scale_invariant_gradient borders,
depth_to_normals, median3x3_downsample's NaN order; the TensorFlow layer forms the nets are made of (padded convs, both
transposed-conv forms, the dense flatten order, 'same' padding of the v2 model); the five nets of networks_original.py and the
three of v2/networks.py on seeded weights; the variable names / shapes TensorFlow really creates; and a checkpoint written by
tf.train.Saver (what examples/example.py:82-83 restores) for demon_amd/tf_checkpoint.py to read.

Run it ONCE in the reference's environment (Dockerfile:14-27 of the reference: python3.5, tensorflow-gpu 1.4, lmbspecialops
built and on LMBSPECIALOPS_LIB), from the root of THIS repository:

    python tools/dump_reference_goldens.py --reference /path/to/demon [--outdir tests/golden/tf] [--full-checkpoint DIR]

and commit tests/golden/tf/ (a few MB).  tests/test_tf_goldens.py consumes the files when they are there and skips otherwise.
Only numpy, json and tools/golden_common.py are needed besides the reference stack; the file is Python-3.5 clean.

    python tools/dump_reference_goldens.py --self-test DIR

writes the same files with THIS repo's CPU oracle standing in for TensorFlow / lmbspecialops (manifest: backend =
"oracle-selftest").  That pins nothing -- it exists so that the file formats and every consuming test are exercised here, where
the reference stack cannot be installed.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import golden_common as G  # noqa: E402

SEED = 1


# =====================================================================================================================================
# the reference stack
# =====================================================================================================================================
class TFBackend(object):
    name = "tensorflow+lmbspecialops"

    def __init__(self, reference):
        sys.path.insert(0, os.path.join(reference, "python"))
        import tensorflow as tf
        import lmbspecialops as sops
        self.tf, self.sops, self.reference = tf, sops, reference
        self.gpu = bool(tf.test.is_gpu_available(True))
        self.data_format = "channels_first" if self.gpu else "channels_last"   # TensorFlow's CPU convs are NHWC only

    def versions(self):
        return {"tensorflow": self.tf.__version__, "numpy": np.__version__, "gpu": self.gpu, "data_format": self.data_format,
                "lmbspecialops": getattr(self.sops, "__version__", os.environ.get("LMBSPECIALOPS_LIB", "unknown"))}

    # ---- ops: lmbspecialops is NCHW on every device
    def run_op(self, case):
        tf = self.tf
        with tf.Graph().as_default(), tf.Session() as session:
            args = [tf.constant(np.asarray(arr, np.float32)) for _, arr in case["inputs"]]
            return session.run(getattr(self.sops, case["op"])(*args, **case["kwargs"]))

    # ---- layers
    def _to_tf(self, x):
        return x if self.data_format == "channels_first" or x.ndim != 4 else x.transpose(0, 2, 3, 1)

    def _from_tf(self, y):
        return y if self.data_format == "channels_first" or y.ndim != 4 else y.transpose(0, 3, 1, 2)

    def run_layer(self, case):
        tf = self.tf
        from depthmotionnet import helpers as H1
        from depthmotionnet import blocks_original as B1
        from depthmotionnet.v2 import helpers as H2
        df, p, kind = self.data_format, case["params"], case["kind"]
        with tf.Graph().as_default(), tf.Session() as session:
            x = tf.constant(self._to_tf(case["x"]))
            with tf.variable_scope("L"):
                if kind == "convrelu2":
                    y = H1.convrelu2_caffe_padding(x, p["num_outputs"], p["kernel_size"], "conv", p["stride"], df)
                elif kind == "conv":
                    f = H1.convrelu_caffe_padding if p["activation"] else H1.conv2d_caffe_padding
                    y = f(x, p["num_outputs"], p["kernel_size"], df, strides=p["strides"], name="conv")
                elif kind == "conv_same":
                    y = H2.convrelu(x, p["num_outputs"], p["kernel_size"], df, strides=p["strides"], name="conv")
                elif kind == "convrelu2_same":
                    y = H2.convrelu2(x, p["num_outputs"], p["kernel_size"], "conv", p["stride"], df)
                elif kind == "upsample_prediction":
                    y = B1._upsample_prediction(x, p["num_outputs"], data_format=df)
                elif kind == "refine":
                    up = case.get("upsampled_prediction")
                    y = B1._refine_caffe_padding(x, p["num_outputs"], df,
                                                 upsampled_prediction=None if up is None else tf.constant(self._to_tf(up)),
                                                 features_direct=tf.constant(self._to_tf(case["features_direct"])))
                elif kind == "flatten_dense":
                    # blocks_original.py:388-396: the motion head flattens the NCHW tensor (converted first in channels_last mode)
                    xin = x if df == "channels_first" else H1.convert_NHWC_to_NCHW(x)
                    y = tf.layers.dense(name="fc", inputs=tf.contrib.layers.flatten(xin), units=p["units"],
                                        activation=H1.myLeakyRelu if p["activation"] else None,
                                        kernel_initializer=H1.default_weights_initializer())
                else:
                    raise ValueError(kind)
            variables = [(v.name.split(":")[0], tuple(v.get_shape().as_list())) for v in tf.global_variables()]
            weights = G.seeded_weights(variables, SEED + 100, head_scale=1.0)
            for v in tf.global_variables():
                v.load(weights[v.name.split(":")[0]], session)
            return self._from_tf(session.run(y)), variables

    # ---- nets
    def _prepare_sculpture(self):
        """the reference's own prepare_input_data (examples/example.py:15-42) on its own PNGs"""
        import ast
        from PIL import Image
        path = os.path.join(self.reference, "examples", "example.py")
        with open(path) as f:
            tree = ast.parse(f.read())
        fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_input_data"][0]
        ns = {"np": np}
        try:
            module = ast.Module(body=[fn], type_ignores=[])
        except TypeError:          # Python < 3.8
            module = ast.Module(body=[fn])
        exec(compile(module, path, "exec"), ns)
        img1 = Image.open(os.path.join(self.reference, "examples", "sculpture1.png"))
        img2 = Image.open(os.path.join(self.reference, "examples", "sculpture2.png"))
        d = ns["prepare_input_data"](img1, img2, "channels_first")
        return d["image_pair"], d["image2_2"]

    def run_nets(self, version, inputs, checkpoint_prefix=None, full_checkpoint=None):
        """-> (variables, {case: {stage key: array}}): bootstrap, 3 x iterative, refinement on seeded weights, every eval() kept"""
        tf = self.tf
        if version == 2 and not self.gpu:
            return None, None          # v2/networks.py builds channels_first graphs only
        with tf.Graph().as_default():
            session = tf.Session(config=tf.ConfigProto(allow_soft_placement=True))
            if version == 1:
                from depthmotionnet.networks_original import BootstrapNet, IterativeNet, RefinementNet
                df = self.data_format
                boot, it, ref = BootstrapNet(session, df), IterativeNet(session, df), RefinementNet(session, df)
            else:
                from depthmotionnet.v2.networks import BootstrapNet, IterativeNet, RefinementNet
                df = "channels_first"
                boot, it, ref = BootstrapNet(session), IterativeNet(session), RefinementNet(session)
            variables = [(v.name.split(":")[0], tuple(v.get_shape().as_list())) for v in tf.global_variables()]
            weights = G.seeded_weights(variables, SEED, consistent_flow=(version == 2))
            for v in tf.global_variables():
                v.load(weights[v.name.split(":")[0]], session)

            def to(a):
                return a if df == "channels_first" or a.ndim != 4 else a.transpose(0, 2, 3, 1)

            def back(d):
                return dict((k, (v if df == "channels_first" or v.ndim != 4 else v.transpose(0, 3, 1, 2))) for k, v in d.items())

            results = {}
            for case, (pair, img2_2) in inputs:
                out = {}
                r = back(boot.eval(to(pair), to(img2_2)))
                out.update(("bootstrap/" + k, v) for k, v in r.items())
                for i in range(3):
                    r = back(it.eval(to(pair), to(img2_2), to(r["predict_depth2"]), to(r["predict_normal2"]), r["predict_rotation"], r["predict_translation"]))
                    out.update(("iterative%d/%s" % (i, k), v) for k, v in r.items())
                if version == 1:
                    q = back(ref.eval(to(pair[:, 0:3]), to(r["predict_depth2"])))
                else:
                    q = back(ref.eval(to(pair[:, 0:3]), to(r["predict_depth2"]), to(r["predict_normal2"])))
                out.update(("refine/" + k, v) for k, v in q.items())
                results[case] = out
            if checkpoint_prefix:   # the bundle format tf.train.Saver writes (examples/example.py:82-83 restores one)
                small = [v for v in tf.global_variables() if v.name.startswith("netRefine/")]
                tf.train.Saver(var_list=small).save(session, checkpoint_prefix, write_meta_graph=False)
            if full_checkpoint:
                tf.train.Saver().save(session, full_checkpoint, write_meta_graph=False)
            session.close()
        return variables, results


# =====================================================================================================================================
# self-test backend: this repo's CPU oracle in the reference's place (pins nothing)
# =====================================================================================================================================
class OracleBackend(object):
    name = "oracle-selftest"

    def __init__(self):
        sys.path.insert(0, ROOT)
        from oracle import net_ref, ops_ref
        from demon_amd import weights as W
        self.ops, self.net, self.W = ops_ref, net_ref, W
        self.gpu = True

    def versions(self):
        return {"numpy": np.__version__, "note": "NOT the reference: written by oracle/ for format and plumbing tests"}

    def run_op(self, case):
        return getattr(self.ops, case["op"])(*[a for _, a in case["inputs"]], **case["kwargs"])

    def run_layer(self, case):
        variables = G_layer_variables(case)
        w = G.seeded_weights(variables, SEED + 100, head_scale=1.0)
        return apply_layer(self.ops, case, w), variables

    def run_nets(self, version, inputs, checkpoint_prefix=None, full_checkpoint=None):
        variables = sorted(self.W.variable_shapes(version=version).items())
        w = G.seeded_weights(variables, SEED, consistent_flow=(version == 2))
        net = self.net.DemonRef(w) if version == 1 else self.net.DemonRefV2(w)
        results = {}
        for case, (pair, img2_2) in inputs:
            out = {}
            r = net.bootstrap(pair, img2_2)
            out.update(("bootstrap/" + k, r[k]) for k in NET_KEYS)
            for i in range(3):
                r = net.iterative(pair, img2_2, r["predict_depth2"], r["predict_normal2"], r["predict_rotation"], r["predict_translation"])
                out.update(("iterative%d/%s" % (i, k), r[k]) for k in NET_KEYS)
            q = net.refine(pair[:, 0:3], r["predict_depth2"])
            out["refine/predict_depth0"] = q["predict_depth0"]
            if version == 2:
                out["refine/predict_normal0"] = q["predict_normal0"]
            results[case] = dict((k, np.asarray(v, np.float32)) for k, v in out.items())
        if checkpoint_prefix:
            from demon_amd import tf_checkpoint as ck
            ck.save_tf_checkpoint(checkpoint_prefix, dict((k, v) for k, v in w.items() if k.startswith("netRefine/")))
        return variables, results


NET_KEYS = ("predict_flow5", "predict_flow2", "predict_depth2", "predict_normal2", "predict_rotation", "predict_translation")


def G_layer_variables(case):
    """the variables tf.layers creates for a layer case under scope "L" (what TFBackend.run_layer asserts it got)"""
    p, kind, cin = case["params"], case["kind"], case["x"].shape[1]
    if kind in ("convrelu2", "convrelu2_same"):
        no = p["num_outputs"]
        cy, cx = (no if isinstance(no, (list, tuple)) else (no, no))
        k = p["kernel_size"]
        return [("L/convy/kernel", (k, 1, cin, cy)), ("L/convy/bias", (cy,)), ("L/convx/kernel", (1, k, cy, cx)), ("L/convx/bias", (cx,))]
    if kind in ("conv", "conv_same"):
        k = p["kernel_size"]
        return [("L/conv/kernel", (k, k, cin, p["num_outputs"])), ("L/conv/bias", (p["num_outputs"],))]
    if kind in ("upsample_prediction", "refine"):
        return [("L/upconv/kernel", (4, 4, p["num_outputs"], cin)), ("L/upconv/bias", (p["num_outputs"],))]
    if kind == "flatten_dense":
        return [("L/fc/kernel", (int(np.prod(case["x"].shape[1:])), p["units"])), ("L/fc/bias", (p["units"],))]
    raise ValueError(kind)


def apply_layer(ops, case, w):
    """a layer case through an implementation with the interface of oracle/ops_ref.py (conv2d_hwio, conv2d_hwio_same,
    deconv4x4s2_crop, dense); tests/test_tf_goldens.py runs the HIP layer entry points through the same function"""
    p, kind, x = case["params"], case["kind"], case["x"]
    if kind == "convrelu2":
        k, s = p["kernel_size"], p["stride"]
        y = ops.conv2d_hwio(x, w["L/convy/kernel"], w["L/convy/bias"], (s, 1), (k // 2, 0), True)
        return ops.conv2d_hwio(y, w["L/convx/kernel"], w["L/convx/bias"], (1, s), (0, k // 2), True)
    if kind == "conv":
        k, s = p["kernel_size"], p["strides"]
        return ops.conv2d_hwio(x, w["L/conv/kernel"], w["L/conv/bias"], (s, s), (k // 2, k // 2), p["activation"])
    if kind == "conv_same":
        s = p["strides"]
        return ops.conv2d_hwio_same(x, w["L/conv/kernel"], w["L/conv/bias"], (s, s), True)
    if kind == "convrelu2_same":
        s = p["stride"]
        y = ops.conv2d_hwio_same(x, w["L/convy/kernel"], w["L/convy/bias"], (s, 1), True)
        return ops.conv2d_hwio_same(y, w["L/convx/kernel"], w["L/convx/bias"], (1, s), True)
    if kind == "upsample_prediction":
        return ops.deconv4x4s2_crop(x, w["L/upconv/kernel"], w["L/upconv/bias"], False)
    if kind == "refine":
        up = ops.deconv4x4s2_crop(x, w["L/upconv/kernel"], w["L/upconv/bias"], True)
        parts = [up, case["features_direct"]] + ([] if case.get("upsampled_prediction") is None else [case["upsampled_prediction"]])
        return np.concatenate(parts, axis=1)
    if kind == "flatten_dense":
        return ops.dense(x.reshape(x.shape[0], -1), w["L/fc/kernel"], w["L/fc/bias"], p["activation"])
    raise ValueError(kind)


# =====================================================================================================================================
def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", help="checkout of lmb-freiburg/demon (its python/ goes on sys.path)")
    ap.add_argument("--outdir", default=os.path.join(ROOT, "tests", "golden", "tf"))
    ap.add_argument("--full-checkpoint", default="", help="also save ALL variables of the original model with tf.train.Saver to this prefix (183 MB, do not commit)")
    ap.add_argument("--self-test", metavar="DIR", default="", help="write the files with this repo's oracle instead (plumbing test, pins nothing)")
    ap.add_argument("--skip-nets", action="store_true")
    args = ap.parse_args()
    if args.self_test:
        backend, outdir = OracleBackend(), args.self_test
    else:
        if not args.reference:
            ap.error("--reference (or --self-test) is required")
        backend, outdir = TFBackend(args.reference), args.outdir
    if not os.path.isdir(outdir):
        os.makedirs(outdir)
    files = []

    cases = G.op_cases()
    outs = []
    for c in cases:
        outs.append(np.asarray(backend.run_op(c), np.float32))
    np.savez_compressed(os.path.join(outdir, "ops.npz"), **G.pack_cases(cases, outs))
    files.append("ops.npz")
    print("ops.npz: %d cases" % len(cases))

    lcases = G.layer_cases()
    blob = {"format_version": np.array(G.FORMAT_VERSION), "count": np.array(len(lcases))}
    for i, c in enumerate(lcases):
        y, variables = backend.run_layer(c)
        want = G_layer_variables(c)
        if sorted((n, tuple(s)) for n, s in variables) != sorted(want):
            raise SystemExit("layer case %s: TensorFlow created %s, expected %s" % (c["tag"], sorted(variables), sorted(want)))
        blob["%d/tag" % i] = np.array(c["tag"])
        blob["%d/variables" % i] = np.array(json.dumps(sorted((n, list(s)) for n, s in variables)))
        blob["%d/out" % i] = np.asarray(y, np.float32)
    np.savez_compressed(os.path.join(outdir, "layers.npz"), **blob)
    files.append("layers.npz")
    print("layers.npz: %d cases" % len(lcases))

    if not args.skip_nets:
        inputs = [("synthetic0", G.synthetic_pair(1, 0))]
        if hasattr(backend, "_prepare_sculpture"):
            inputs.append(("sculpture", backend._prepare_sculpture()))
        for version in (1, 2):
            ck = os.path.join(outdir, "ckpt", "netRefine_seed%d" % SEED) if version == 1 else None
            if ck and not os.path.isdir(os.path.dirname(ck)):
                os.makedirs(os.path.dirname(ck))
            variables, results = backend.run_nets(version, inputs, checkpoint_prefix=ck, full_checkpoint=(args.full_checkpoint or None) if version == 1 else None)
            if variables is None:
                print("version %d nets skipped (need a GPU build of TensorFlow: channels_first only)" % version)
                continue
            blob = {"format_version": np.array(G.FORMAT_VERSION), "seed": np.array(SEED),
                    "variables": np.array(json.dumps(sorted((n, list(s)) for n, s in variables))),
                    "cases": np.array(json.dumps([c for c, _ in inputs]))}
            for case, (pair, img2_2) in inputs:
                blob["%s/in/image_pair" % case] = pair
                blob["%s/in/image2_2" % case] = img2_2
                for k, v in results[case].items():
                    blob["%s/out/%s" % (case, k)] = np.asarray(v, np.float32)
            name = "nets_original.npz" if version == 1 else "nets_v2.npz"
            np.savez_compressed(os.path.join(outdir, name), **blob)
            files.append(name)
            print("%s: %d variables, cases %s" % (name, len(variables), [c for c, _ in inputs]))
            if ck:
                files.append("ckpt/" + os.path.basename(ck) + ".index")

    manifest = {"backend": backend.name, "versions": backend.versions(), "seed": SEED, "format_version": G.FORMAT_VERSION,
                "files": files, "written": time.strftime("%Y-%m-%d %H:%M:%S")}
    with open(os.path.join(outdir, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", outdir)


if __name__ == "__main__":
    main()
