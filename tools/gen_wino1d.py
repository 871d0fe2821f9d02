#!/usr/bin/env python3
"""Generates demon_amd/csrc/wino1d_tables.h: the transforms of the 1-D minimal-filtering convolutions of conv_wino.hip.

Two consecutive outputs of a k-tap filter along one axis:
  stride 1, k = 3:  F(2,3) -- 4 products instead of 6
  stride 2, k taps: the even / odd input samples see the even / odd taps as two stride-1 filters with re = ceil(k/2) and
                    ro = floor(k/2) taps (polyphase split); F(2,re) + F(2,ro) = k + 2 products instead of 2k
The matrices are the Toom-Cook construction (Lavin & Gray, "Fast algorithms for convolutional neural networks", wincnn) in exact
rationals; every row's denominators are moved from the weight transform G into the output transform AT, so that G and BT have small
integer entries.  The script checks the identity AT [(G g) . (BT d)] = correlation(d, g) exactly before it writes anything.

  python tools/gen_wino1d.py            (re)writes the header
"""
import os
from fractions import Fraction as Fr
from math import lcm

POINTS = {2: [0, -1], 3: [0, 1, -1], 4: [0, 1, -1, 2], 5: [0, 1, -1, 2, -2], 1: []}   # F(2,r): r finite points (+ infinity)
POINTS_BY_COUNT = {5: [0, 1, -1, 2, -2], 4: [0, 1, -1, 2]}                               # F(4,3) / F(4,2): m + r - 2 finite points


def toom(m, r):
    """AT (m x a), G (a x r), BT (a x a), a = m + r - 1"""
    a = m + r - 1
    if r == 1:   # F(m,1): m outputs, one tap: o_k = d_k g
        eye = [[Fr(1) if i == j else Fr(0) for j in range(m)] for i in range(m)]
        return eye, [[Fr(1)] for _ in range(m)], [row[:] for row in eye]
    pts = [Fr(p) for p in (POINTS[r] if m == 2 else POINTS_BY_COUNT[a - 1])]
    assert len(pts) == a - 1

    def polymul(p, q):
        res = [Fr(0)] * (len(p) + len(q) - 1)
        for i, x in enumerate(p):
            for j, y in enumerate(q):
                res[i + j] += x * y
        return res
    M = [Fr(1)]
    for p in pts:
        M = polymul(M, [-p, Fr(1)])
    Ni = []
    for i, p in enumerate(pts):
        v = Fr(1)
        for j, q in enumerate(pts):
            if j != i:
                v *= (p - q)
        Ni.append(v)
    AT = [[pts[i] ** k for i in range(a - 1)] + [Fr(1) if k == m - 1 else Fr(0)] for k in range(m)]
    G = [[pts[i] ** k / Ni[i] for k in range(r)] for i in range(a - 1)] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    BT = []
    for p in pts:
        carry, out = Fr(0), []
        for c in M[::-1][:-1]:
            carry = c + carry * p
            out.append(carry)
        q = out[::-1]
        BT.append(q + [Fr(0)] * (a - len(q)))
    BT.append(M + [Fr(0)] * (a - len(M)))
    return AT, G, BT


def kind_matrices(taps, stride):
    """full matrices over the window: AT (2 x NUV), G (NUV x taps), BT (NUV x WIN)"""
    if stride == 1:
        AT, G, BT = toom(2, taps)
        return AT, G, BT, taps + 1
    re, ro = (taps + 1) // 2, taps // 2
    win = taps + 2
    ATe, Ge, BTe = toom(2, re)
    ATo, Go, BTo = toom(2, ro)
    ne, no = len(Ge), len(Go)
    AT = [ATe[k] + ATo[k] for k in range(2)]
    G = [[Fr(0)] * taps for _ in range(ne + no)]
    BT = [[Fr(0)] * win for _ in range(ne + no)]
    for i in range(ne):
        for b in range(re):
            G[i][2 * b] = Ge[i][b]
        for a_ in range(re + 1):
            BT[i][2 * a_] = BTe[i][a_]
    for i in range(no):
        for b in range(ro):
            G[ne + i][2 * b + 1] = Go[i][b]
        for a_ in range(ro + 1):
            BT[ne + i][2 * a_ + 1] = BTo[i][a_]
    return AT, G, BT, win


def kind_matrices4(taps, stride):
    """FOUR outputs per window: AT (4 x NUV), G (NUV x taps), BT (NUV x WIN); stride 2 = polyphase F(4,re) + F(4,ro)"""
    if stride == 1:
        AT, G, BT = toom(4, taps)
        return AT, G, BT, taps + 3
    re, ro = (taps + 1) // 2, taps // 2
    win = 2 * 4 + taps - 2
    ATe, Ge, BTe = toom(4, re)
    ATo, Go, BTo = toom(4, ro)
    ne, no = len(Ge), len(Go)
    AT = [ATe[k] + ATo[k] for k in range(4)]
    G = [[Fr(0)] * taps for _ in range(ne + no)]
    BT = [[Fr(0)] * win for _ in range(ne + no)]
    for i in range(ne):
        for b in range(re):
            G[i][2 * b] = Ge[i][b]
        for a_ in range(re + 3):
            BT[i][2 * a_] = BTe[i][a_]
    for i in range(no):
        for b in range(ro):
            G[ne + i][2 * b + 1] = Go[i][b]
        for a_ in range(ro + 3):
            BT[ne + i][2 * a_ + 1] = BTo[i][a_]
    return AT, G, BT, win


def render4(name, taps, stride, comment):
    AT, G, BT, win = kind_matrices4(taps, stride)
    AT, G = normalise(AT, G)
    check(AT, G, BT, taps, stride, win)
    nuv = len(G)
    out = ["// " + comment, "struct %s {" % name,
           "    static constexpr int TAPS = %d, STRIDE = %d, NUV = %d, WIN = %d, OUT = 4;" % (taps, stride, nuv, win),
           "    static __host__ __device__ __forceinline__ float g(int e, int t) {",
           "        constexpr float G[NUV][TAPS] = {%s};" % ", ".join("{" + ", ".join(cf(v) for v in row) + "}" for row in G),
           "        return G[e][t];", "    }",
           "    static __device__ __forceinline__ void input(const float (&d)[WIN], float (&t)[NUV]) {", emit_input(BT, win, name), "    }"]
    names = ["m[%d]" % e for e in range(nuv)]
    out.append("    // (T = float, or a vector of channels: a lane's accumulator registers ARE four consecutive channels, and on float4 the compiler emits")
    out.append("    // packed fp32 instructions -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, two channels each)")
    out.append("    template <class T> static __device__ __forceinline__ void output(const T (&m)[NUV], T (&o)[OUT]) {")
    for k in range(4):
        out.append("        o[%d] = %s;" % (k, linear(AT[k], names)))
    out += ["    }", "};", ""]
    return out


def normalise(AT, G):
    """integer G: row e times its common denominator, column e of AT divided by it"""
    AT = [row[:] for row in AT]
    G = [row[:] for row in G]
    for e, row in enumerate(G):
        den = 1
        for v in row:
            den = lcm(den, v.denominator)
        G[e] = [v * den for v in row]
        for k in range(len(AT)):
            AT[k][e] = AT[k][e] / den
    return AT, G


def check(AT, G, BT, taps, stride, win):
    import random
    rnd = random.Random(1)
    for _ in range(20):
        d = [Fr(rnd.randint(-9, 9)) for _ in range(win)]
        g = [Fr(rnd.randint(-9, 9)) for _ in range(taps)]
        U = [sum(G[e][t] * g[t] for t in range(taps)) for e in range(len(G))]
        T = [sum(BT[e][n] * d[n] for n in range(win)) for e in range(len(G))]
        Y = [sum(AT[k][e] * U[e] * T[e] for e in range(len(G))) for k in range(len(AT))]
        ref = [sum(d[stride * k + t] * g[t] for t in range(taps)) for k in range(len(AT))]
        assert Y == ref, (taps, stride, Y, ref)


def cf(v):
    """C float literal of a rational"""
    if v.denominator == 1:
        return "%d.0f" % v.numerator
    return "(%d.0f / %d.0f)" % (v.numerator, v.denominator)


def term(c, name, first):
    """c * name as a C expression piece with its sign"""
    sign = "-" if c < 0 else ("" if first else "+")
    mag = abs(c)
    body = name if mag == 1 else "%s * %s" % (cf(mag), name)
    return ("%s%s" % (sign, body)) if first else (" %s %s" % (sign, body))


def linear(coeffs, names):
    parts, first = [], True
    for c, n in zip(coeffs, names):
        if c == 0:
            continue
        parts.append(term(c, n, first))
        first = False
    return "".join(parts) if parts else "0.0f"


def emit_input(BT, win, tag):
    """t[e] = BT[e] . d with the rows of the point pairs +-p sharing their even / odd parts"""
    lines, done, ops = [], set(), 0
    names = ["d[%d]" % n for n in range(win)]
    for i in range(len(BT)):
        if i in done:
            continue
        pair = None
        for j in range(i + 1, len(BT)):
            if j in done:
                continue
            if all(abs(BT[i][n]) == abs(BT[j][n]) for n in range(win)) and any(BT[i][n] != 0 and BT[i][n] == -BT[j][n] for n in range(win)) \
                    and any(BT[i][n] != 0 and BT[i][n] == BT[j][n] for n in range(win)):
                pair = j
                break
        if pair is None:
            lines.append("    t[%d] = %s;" % (i, linear(BT[i], names)))
            continue
        j = pair
        same = [BT[i][n] if BT[i][n] == BT[j][n] else Fr(0) for n in range(win)]
        diff = [BT[i][n] if BT[i][n] == -BT[j][n] else Fr(0) for n in range(win)]
        lines.append("    { const float s = %s, a = %s; t[%d] = s + a; t[%d] = s - a; }" % (linear(same, names), linear(diff, names), i, j))
        done.add(j)
    return "\n".join(lines)


KINDS = [(3, 1), (5, 2), (7, 2), (9, 2)]


def render():
    """the text of wino1d_tables.h (every matrix checked in exact rationals on the way)"""
    out = ["// wino1d_tables.h -- GENERATED by tools/gen_wino1d.py (do not edit): transforms of the 1-D minimal-filtering convolutions",
           "// of conv_wino.hip.  kind 0: 3 taps stride 1 (F(2,3)); kinds 1 / 2 / 3: 5 / 7 / 9 taps stride 2 (polyphase F(2,re) + F(2,ro)).",
           "#pragma once", "", "namespace demon {", "",
           "template <int KIND> struct Wino1D;", ""]
    for kind, (taps, stride) in enumerate(KINDS):
        AT, G, BT, win = kind_matrices(taps, stride)
        AT, G = normalise(AT, G)
        check(AT, G, BT, taps, stride, win)
        nuv = len(G)
        out.append("// %d taps, stride %d: %d products per 2 outputs instead of %d; window of %d inputs" % (taps, stride, nuv, 2 * taps, win))
        out.append("template <> struct Wino1D<%d> {" % kind)
        out.append("    static constexpr int TAPS = %d, STRIDE = %d, NUV = %d, WIN = %d;" % (taps, stride, nuv, win))
        out.append("    // weight transform (integer entries; the denominators sit in the output transform): U[e] = sum_t G[e][t] w[t]")
        out.append("    static __host__ __device__ __forceinline__ float g(int e, int t) {")
        out.append("        constexpr float G[NUV][TAPS] = {%s};" % ", ".join("{" + ", ".join(cf(v) for v in row) + "}" for row in G))
        out.append("        return G[e][t];")
        out.append("    }")
        out.append("    // input transform: t[e] = sum_n BT[e][n] d[n]")
        out.append("    static __device__ __forceinline__ void input(const float (&d)[WIN], float (&t)[NUV]) {")
        out.append(emit_input(BT, win, kind))
        out.append("    }")
        out.append("    // output transform: o[k] = sum_e AT[k][e] m[e]")
        names = ["m[%d]" % e for e in range(nuv)]
        out.append("    template <class T> static __device__ __forceinline__ void output(const T (&m)[NUV], T &o0, T &o1) {")
        out.append("        o0 = %s;" % linear(AT[0], names))
        out.append("        o1 = %s;" % linear(AT[1], names))
        out.append("    }")
        out.append("};")
        out.append("")
    # four outputs per window (conv_wino3.hip, conv_wino4.hip)
    out += render4("Wino43", 3, 1, "3 taps, stride 1, FOUR outputs per window of 6 inputs: 6 products instead of 12 (points 0, +-1, +-2, infinity)")
    out += render4("Wino4K5S2", 5, 2, "5 taps, stride 2, FOUR outputs per window of 11 inputs: polyphase F(4,3) + F(4,2) = 11 products instead of 20")
    out += render4("Wino4K3S2", 3, 2, "3 taps, stride 2, FOUR outputs per window of 9 inputs: polyphase F(4,2) + F(4,1) = 9 products instead of 12")
    out.append("}  // namespace demon")
    return "\n".join(out) + "\n"


HEADER = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "demon_amd", "csrc", "wino1d_tables.h"))


def main():
    text = render()
    with open(HEADER, "w") as f:
        f.write(text)
    print("wrote", HEADER)


if __name__ == "__main__":
    main()
