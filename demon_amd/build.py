"""Builds libdemon_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdemon_hip.so")
SOURCES = ["demon_api.hip", "conv_mfma.hip", "conv_patch.hip", "conv_small.hip", "conv_pair.hip", "conv_stream.hip", "conv_frag.hip", "conv_wino.hip", "conv_wino3.hip", "conv_wino4.hip", "dense_stream.hip", "conv_thin.hip", "conv_row.hip", "ops.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def csrc_sha():
    """sha1 over the kernel sources (every .hip except the host-side demon_api.hip, plus internal.h) and the shipped launch
    plans: stamps profiles (tools/pmc_summary.py) so that bench.py can tell a counter summary measured on these kernels and
    plans from a stale one"""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = [os.path.join(CSRC, n) for n in sorted(SOURCES) if n != "demon_api.hip"] + [os.path.join(CSRC, "internal.h"), os.path.join(CSRC, "wino1d_tables.h")]
    files += sorted(glob.glob(os.path.join(HERE, "tuned", "*.json")))
    for path in files:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _needs_build(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), tag=""):
    """tag / extra_flags: diagnostic variants (e.g. tag="tl", extra_flags=["-DDEMON_TIMELINE"] -> libdemon_hip_tl.so with its own
    object files); the product library is the untagged build"""
    hipcc = _hipcc()
    out = OUT if not tag else OUT.replace(".so", "_%s.so" % tag)
    headers = [os.path.join(CSRC, "internal.h"), os.path.join(CSRC, "wino1d_tables.h"), os.path.join(HERE, "..", "include", "demon_hip.h")]
    objs = []
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", (".%s.o" % tag) if tag else ".o"))
        objs.append(o)
        if force or _needs_build(o, [s] + headers):
            # conv_wino.hip: the SLP vectoriser packs the +-1 transforms into v_pk_add_f32 plus register shuffles -- more VALU issue
            # slots beside the fp32 MFMAs, which share the SIMD with the vector ALU
            per_file = ["-fno-slp-vectorize"] if src in ("conv_wino.hip", "conv_wino3.hip", "conv_wino4.hip", "conv_row.hip") else []
            jobs.append([hipcc] + FLAGS + per_file + list(extra_flags) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _needs_build(out, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    if "--sets4" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, extra_flags=["-DDEMON_STREAM_SETS=4"], tag="s4"))
    elif "--sets2" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, extra_flags=["-DDEMON_STREAM_SETS=2"], tag="s2"))
    elif "--dbg" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, extra_flags=["-DDEMON_STREAM_DBG"], tag="dbg"))
    elif "--unbounded" in sys.argv:   # round 4's 1 GiB buffer descriptors instead of the true tensor extents: A/B timing of the range check only
        print(build(force="--force" in sys.argv, verbose=True, extra_flags=["-DDEMON_RSRC_UNBOUNDED"], tag="unb"))
    elif "--timeline" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, extra_flags=["-DDEMON_TIMELINE"], tag="tl"))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
