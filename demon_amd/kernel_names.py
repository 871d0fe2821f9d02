"""Two names for one kernel instance: the tag demon_profile_full() reports (e.g. `conv_frag<128x32,v6>+splitk`) and the
demangled template name rocprofv3 --kernel-trace prints.  bench.py and tools/pmc_summary.py join their tables on these."""
import functools
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


@functools.lru_cache(None)
def frag_variants():
    """variant index -> (WM, WN, TM, TN, KS, KW), read from the dispatch switch in csrc/conv_frag.hip"""
    src = open(os.path.join(CSRC, "conv_frag.hip")).read()
    table = {}
    for m in re.finditer(r"case (\d+): launch_frag_instance<([\d, ]+)>", src):
        table[int(m.group(1))] = tuple(int(x) for x in m.group(2).split(","))
    m = re.search(r"default: launch_frag_instance<([\d, ]+)>", src)
    if m:
        table[max(table) + 1] = tuple(int(x) for x in m.group(1).split(","))
    return table


def kernel_tag(name):
    """rocprofv3 kernel name -> profile tag of the same template instance (without the +splitk suffix)"""
    m = re.search(r"conv_mfma_kernel<(\d+), (\d+)", name)
    if m:
        return "conv_mfma<%sx%s>" % m.groups()
    m = re.search(r"conv_patch_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if m:
        bm, wm, wn, tm, tn, taps = map(int, m.groups())
        return "conv_patch<%dx%d,t%d>" % (bm, wn * tn * (16 if bm == 16 else 32), taps)
    m = re.search(r"conv_stream_kernel<(\d+), (\d+), (\d+), (\d+)", name)
    if m:
        nw, tm, tn, kw = map(int, m.groups())
        return "conv_stream<%dx%d,w%dk%d>" % (32 * nw * tm, 32 * tn, nw, kw)
    m = re.search(r"conv_frag_chain_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if m:
        params = tuple(map(int, m.groups())) + (1,)
        wm, wn, tm, tn = params[:4]
        for v, p in sorted(frag_variants().items()):
            if p == params:
                return "conv_frag_chain<%dx%d,v%d>" % (32 * tm * wm, 32 * tn * wn, v)
        return "conv_frag_chain<%dx%d,?>" % (32 * tm * wm, 32 * tn * wn)
    m = re.search(r"conv_stream_chain_kernel<(\d+), (\d+), (\d+)", name)
    if m:
        nw, tm, tn = map(int, m.groups())
        return "conv_stream_chain<%dx%d,w%d>" % (32 * nw * tm, 32 * tn, nw)
    m = re.search(r"conv_frag_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if m:
        params = tuple(map(int, m.groups()))
        wm, wn, tm, tn, ks, kw = params
        for v, p in sorted(frag_variants().items()):
            if p == params:
                return "conv_frag<%dx%d,v%d>" % (32 * tm * wm, 32 * tn * wn, v)
        return "conv_frag<%dx%d,?>" % (32 * tm * wm, 32 * tn * wn)
    m = re.search(r"wino_deconv_kernel<(\d+), \d+, \d+, (\d+), 2>", name)
    if m:   # <TN, EPT, OCC, MB, KH = 2>: the reduction split in two inside the workgroup (round 6, variant 6)
        return "wino_deconv<%dx%d,kh2>" % (16 * int(m.group(2)), 16 * int(m.group(1)))
    m = re.search(r"wino_deconv_kernel<(\d+), \d+, \d+, (\d+)", name)
    if m:   # <tile blocks, staged elements per thread, waves per SIMD, 16-channel blocks per wave>
        return "wino_deconv<%dx%d>" % (16 * int(m.group(2)), 16 * int(m.group(1)))
    m = re.search(r"wino1d_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if m:   # (the 3 x 3 layers, tag "t3x3", run the 3-tap instance: the template name cannot tell them apart)
        kind, axis, wm, wn, tn, kg = map(int, m.groups())
        shapes = {(2, 2, 2): 0, (4, 1, 4): 1, (2, 2, 4): 2, (4, 2, 4): 3}
        wide = {(4, 1, 3, 4): 8, (2, 2, 3, 2): 9, (4, 2, 3, 4): 10, (1, 4, 4, 1): 11, (1, 4, 2, 2): 12}   # 48 / 96 tiles per workgroup; one 16-channel block
        v = wide[(wm, wn, tn, kg)] if (wm, wn, tn, kg) in wide else shapes.get((wm, wn, tn), -1) + 4 * (kg - 1)
        return "wino1d<t%d,v%d>" % (3 if kind == 0 else 3 + 2 * kind, v)
    m = re.search(r"wino3_rows_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (\d+)", name)
    if m:   # <WM, WN, TN, KG, MASK, FORM>; FORM 0: F(2,3) tiles, 1: F(4,3) tiles, 2: the stride-2 form
        shape = tuple(map(int, m.groups()[:4]))
        form = int(m.group(6))
        if form == 2:
            shapes2 = {(4, 1, 2, 1): 16, (2, 2, 2, 1): 17, (4, 2, 2, 1): 18, (8, 1, 2, 1): 19}
            return "wino3rows<s2t3x3,v%d>" % shapes2.get(shape, -1)
        if form == 1:
            shapes4 = {(2, 2, 2, 2): 8, (4, 1, 2, 2): 9, (2, 2, 3, 1): 10, (4, 1, 3, 1): 11, (1, 4, 2, 1): 12, (1, 4, 3, 1): 13, (2, 1, 2, 2): 14, (4, 2, 3, 1): 15}
            return "wino3rows<f4t3x3,v%d>" % shapes4.get(shape, -1)
        shapes3 = {(2, 2, 4, 1): 0, (2, 4, 4, 1): 1, (4, 1, 4, 1): 2, (4, 2, 4, 1): 3, (1, 4, 4, 1): 4, (1, 4, 2, 2): 5, (2, 2, 2, 2): 6, (4, 1, 2, 2): 7}
        return "wino3rows<t3x3,v%d>" % shapes3.get(shape, -1)
    m = re.search(r"wino4_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (true|false)", name)
    if m:   # <KIND, AXIS, WM, WN, TN, KG, MASK, SAMEPAD, PERSIST>; PERSIST: the tile-walking form (round 6)
        kind, axis, wm, wn, tn, kg = map(int, m.groups()[:6])
        shapes4 = {(4, 1, 4, 1): 0, (4, 1, 2, 2): 1, (2, 2, 4, 1): 2, (2, 2, 2, 2): 3, (4, 2, 2, 2): 4, (4, 2, 4, 1): 5, (4, 1, 1, 4): 6, (2, 2, 1, 4): 7, (2, 2, 2, 1): 8,
                   (4, 2, 3, 2): 9, (2, 2, 3, 2): 10, (4, 1, 3, 2): 11, (2, 2, 3, 1): 12, (4, 2, 3, 1): 13}
        return "wino4<t%d,v%d%s>" % (3 if kind == 0 else 5, shapes4.get((wm, wn, tn, kg), -1), ",walk" if m.group(9) == "true" else "")
    m = re.search(r"conv_row_kernel<(\d+), ", name)
    if m:
        return "conv_row<32x128,t%d>" % (3 + 2 * int(m.group(1)))
    m = re.search(r"conv_thin_kernel<(\d+)", name)
    if m:
        return "conv_thin<32x512,t%s>" % m.group(1)
    m = re.search(r"dense_stream_kernel<(true|false)", name)
    if m:
        return "dense_stream<128x32,v%d>" % (1 if m.group(1) == "true" else 0)
    m = re.search(r"deconv4_kernel<(\d+), (\d+), (\d+)", name)
    if m:
        bm, wm, wn = map(int, m.groups())
        return "deconv4<%dx%d>" % (bm, wn * 32)
    m = re.search(r"demon::(\w+?)_kernel", name)
    return m.group(1) if m else name.split("(")[0]


def rocprof_kernel_name(tag):
    """profile tag -> how rocprofv3 --kernel-trace names the kernel (profiles/*_kernel_stats.csv)"""
    base = tag.split("+")[0]
    fam, _, rest = base.partition("<")
    dims = rest.rstrip(">").split(",")[0].split("x") if rest else []
    if fam == "conv_mfma" and len(dims) == 2:
        return "demon::conv_mfma_kernel<%s, %s, ...>" % tuple(dims)
    if fam == "conv_patch" and len(dims) == 2:
        return "demon::conv_patch_kernel<%s, ...> (%sx%s tile, %s taps)" % (dims[0], dims[0], dims[1], rest.rstrip(">").split(",t")[-1])
    if fam == "wino_deconv" and len(dims) == 2:
        kh2 = ",kh2" in rest
        return "demon::wino_deconv_kernel<%d, ..., %d, %d> (%s channels x %s tiles per workgroup%s)" % (
            int(dims[1]) // 16, int(dims[0]) // 16, 2 if kh2 else 1, dims[0], dims[1], ", reduction split in two inside the workgroup" if kh2 else "")
    if fam == "wino4":
        return "demon::wino4_kernel<...> (%s)" % rest.rstrip(">")
    if fam == "wino3rows":
        return "demon::wino3_rows_kernel<...> (%s)" % rest.rstrip(">")
    if fam == "wino1d":
        return "demon::wino1d_kernel<...> (%s)" % rest.rstrip(">")
    if fam == "deconv4":
        return "demon::deconv4_kernel<%s, ...>" % dims[0]
    if fam == "conv_stream" and len(dims) == 2:
        w, k = rest.rstrip(">").split(",")[1].lstrip("w").split("k")
        tm = int(dims[0]) // (32 * int(w))
        return "demon::conv_stream_kernel<%s, %d, %d, %s> (%sx%s tile)" % (w, tm, int(dims[1]) // 32, k, dims[0], dims[1])
    if fam == "conv_stream_chain" and len(dims) == 2:
        w = int(rest.rstrip(">").split(",w")[-1])
        return "demon::conv_stream_chain_kernel<%d, %d, %d> (%sx%s tile)" % (w, int(dims[0]) // (32 * w), int(dims[1]) // 32, dims[0], dims[1])
    if fam == "conv_frag_chain" and len(dims) == 2:
        v = rest.rstrip(">").split(",v")[-1]
        p = frag_variants().get(int(v)) if v.isdigit() else None
        if p:
            return "demon::conv_frag_chain_kernel<%d, %d, %d, %d, %d> (%sx%s tile)" % (p[:5] + (dims[0], dims[1]))
    if fam == "conv_frag" and len(dims) == 2:
        v = rest.rstrip(">").split(",v")[-1]
        p = frag_variants().get(int(v)) if v.isdigit() else None
        if p:
            return "demon::conv_frag_kernel<%d, %d, %d, %d, %d, %d> (%sx%s tile)" % (p + (dims[0], dims[1]))
    return "demon::%s_kernel" % fam
