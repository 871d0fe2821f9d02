"""CPU checks of the minimal-filtering forms conv_wino.hip computes (no GPU, no oracle: exact rational arithmetic and numpy):

  * tools/gen_wino1d.py's Toom-Cook matrices reproduce the correlation EXACTLY (rationals) for every 1-D kind the kernel has, and
    the committed header demon_amd/csrc/wino1d_tables.h is what the generator writes (nobody edited it by hand);
  * the F(2,2) x F(2,2) form of the 4x4 stride-2 transposed conv (blocks_original.py:64-75), written out as the kernel does it
    (tap sums, row-then-column input transform, output sums), equals PyTorch's conv_transpose2d.
The GPU twins (tests/test_variants_gpu.py::test_minimal_filtering_*) hold the kernels themselves to PyTorch at 1e-5."""
import importlib.util
import os
from fractions import Fraction as Fr

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_wino1d", os.path.join(ROOT, "tools", "gen_wino1d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generated_header_is_current_and_exact():
    gen = _gen()
    text = gen.render()   # asserts the exact identity for every kind on the way
    with open(gen.HEADER) as f:
        assert f.read() == text, "demon_amd/csrc/wino1d_tables.h is stale: run python tools/gen_wino1d.py"
    # products per two outputs: F(2,3) = 4; stride 2: taps + 2
    for kind, (taps, stride) in enumerate(gen.KINDS):
        AT, G, BT, win = gen.kind_matrices(taps, stride)
        assert len(G) == (4 if stride == 1 else taps + 2) and win == (4 if stride == 1 else taps + 2)
        AT, G = gen.normalise(AT, G)
        assert all(v.denominator == 1 for row in G for v in row) and all(v.denominator == 1 for row in BT for v in row)
        gen.check(AT, G, BT, taps, stride, win)


def test_toom_cook_matrices_against_random_rationals():
    gen = _gen()
    import random
    rnd = random.Random(7)
    for r in (2, 3, 4, 5):
        AT, G, BT = gen.toom(2, r)
        a = r + 1
        for _ in range(10):
            d = [Fr(rnd.randint(-50, 50), rnd.randint(1, 9)) for _ in range(a)]
            g = [Fr(rnd.randint(-50, 50), rnd.randint(1, 9)) for _ in range(r)]
            U = [sum(G[i][k] * g[k] for k in range(r)) for i in range(a)]
            T = [sum(BT[i][k] * d[k] for k in range(a)) for i in range(a)]
            Y = [sum(AT[k][i] * U[i] * T[i] for i in range(a)) for k in range(2)]
            assert Y == [sum(d[k + t] * g[t] for t in range(r)) for k in range(2)]


def test_deconv_f22_form_equals_conv_transpose():
    """wino_deconv_kernel's arithmetic in numpy: per sub-pixel class the 9 tap sums U[u][v], the 3x3 neighbourhood transformed by
    rows then columns (d0 - d1, d1, d2 - d1), O[a][b] = sum of the four products around (a, b)"""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    N, Cin, Cout, H, W = 2, 5, 3, 5, 7
    x = rng.standard_normal((N, Cin, H, W))
    w = rng.standard_normal((4, 4, Cout, Cin))   # TF layout [kh][kw][Cout][Cin]
    ref = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), stride=2, padding=1).numpy()
    tap_a = [[1, 3], [0, 2]]          # kernel row / column of tap t for output parity p (plan_layer's K table)
    S = [[1], [0, 1], [0]]            # taps summed into U[u]
    out = np.zeros_like(ref)
    xp = np.pad(x, ((0, 0), (0, 0), (1, 2), (1, 2)))
    for py in range(2):
        for px in range(2):
            U = np.zeros((3, 3, Cout, Cin))
            for u in range(3):
                for v in range(3):
                    for ty in S[u]:
                        for tx in S[v]:
                            U[u, v] += w[tap_a[py][ty], tap_a[px][tx]]
            for r in range((H + 1) // 2):
                for c in range((W + 1) // 2):
                    d = xp[:, :, 2 * r + py:2 * r + py + 3, 2 * c + px:2 * c + px + 3]
                    rows = np.stack([d[:, :, 0] - d[:, :, 1], d[:, :, 1], d[:, :, 2] - d[:, :, 1]], 2)
                    t = np.stack([rows[..., 0] - rows[..., 1], rows[..., 1], rows[..., 2] - rows[..., 1]], 3)
                    M = np.einsum("uvoc,ncuv->nouv", U, t)
                    for a in range(2):
                        for b in range(2):
                            y, xx = 2 * r + a, 2 * c + b
                            if y < H and xx < W:
                                out[:, :, 2 * y + py, 2 * xx + px] = M[:, :, a, b] + M[:, :, a, b + 1] + M[:, :, a + 1, b] + M[:, :, a + 1, b + 1]
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-12)
