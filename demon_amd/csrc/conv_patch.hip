// conv_patch.hip -- patch-staged fp32-MFMA convolution for the k x 1 / 1 x k / 3 x 3 convs and the
// 2 x 2 sub-pixel classes of the transposed convs (helpers.py:70-153, blocks_original.py:64-75, :97-110).
//
// Same GEMM view as conv_mfma.hip (A = weights, M = Cout; N = pixels so that NCHW stores coalesce), but the
// B operand is not gathered element by element.  One K-step covers ALL taps of CKS input channels:
//   * the raw input patch of the pixel tile (tile + halo, zero filled outside the image) for those CKS
//     channels is copied to LDS once: CKS*PH*PW floats instead of NTAPS*CKS*BN im2col elements (3-9x fewer
//     loads), with per-thread offsets / validity computed ONCE per kernel -> zero address VALU per load;
//   * every tap reads its shifted view of the patch straight from LDS: lane j (one output pixel) reads
//     patch[ci][y_j*sh + dy][x_j*sw + dx]; consecutive lanes are consecutive x, i.e. conflict free for
//     stride 1 and 2-way for stride 2 (ds_read_b32 half-wave groups).
// Reduction order per output: channel chunk major, then tap, then channel inside the chunk.
#include <type_traits>

#include "internal.h"

namespace demon {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// n / d for the small indices of the prologue / epilogue (n < 2^20, d < 2^12) with the host-provided magic ceil(2^32 / d)
__device__ __forceinline__ int fdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }  // magic 0 <=> d == 1

// BM == 16 (Cout <= 16 heads): v_mfma_f32_16x16x4_f32 -- a wave owns 16 channels x TN*16 pixels and one MFMA eats 4 reduction
// indices (lane: row / pixel = lane & 15, k = lane >> 4); same FLOP rate as the 32x32x2 tile without multiplying 16 rows of zeros.
template <int BM, int WM, int WN, int TM, int TN, int NTAPS, int CKS, int EPT>
__global__ __launch_bounds__(64 * WM * WN) void conv_patch_kernel(PatchArgs a)
{
    constexpr bool M16 = BM == 16;
    constexpr int NT = 64 * WM * WN;
    constexpr int KD = NTAPS * CKS;  // reduction depth of one K-step
    constexpr int KG = M16 ? 4 : 2;  // reduction indices per MFMA
    constexpr int NG = KD / KG;      // MFMA groups per step
    constexpr int PXW = M16 ? 16 : 32;  // pixels per MFMA column block
    constexpr int NR = M16 ? 4 : 16;    // accumulator registers per MFMA
    constexpr int A4 = KD * BM / 4;  // float4 chunks of the A tile
    constexpr int APER = (A4 + NT - 1) / NT;
    static_assert((M16 ? (WM == 1 && TM == 1) : BM == WM * TM * 32) && CKS % KG == 0, "bad tile");
    using AccT = typename std::conditional<M16, floatx4, floatx16>::type;

    TlScope tl(a.tl);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                           // [2][KD][BM]
    float *Ps = smem + 2 * KD * BM;             // [2][patch_floats]
    const int patch_floats = a.G * CKS * a.PS;

    const int tid = threadIdx.x;
    const int cls = blockIdx.z / a.ksplit;
    const int zs = blockIdx.z - cls * a.ksplit;
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
    const int m0 = by * BM;
    // tile -> (image group, tile row, tile col)
    const int tyg = fdiv((int)bx, a.m_tilesx);
    const int tx = (int)bx - tyg * a.tiles_x;
    const int tgrp = fdiv(tyg, a.m_tilesy);
    const int ty = tyg - tgrp * a.tiles_y;
    const int n0 = tgrp * a.G;
    const int y_org = ty * a.TH * a.sh + a.oy0[cls];  // input coordinates of patch element (0,0)
    const int x_org = tx * a.TW * a.sw + a.ox0[cls];
    const float *__restrict__ wp = a.wp + (long)cls * a.cls_w_stride;
    const float *__restrict__ in0 = a.in + (long)n0 * a.in_n_stride;

    // ---- patch loader: element e = tid + i*NT of the [G*CKS][PH*PW] patch, decoded once
    const int plane_elems = a.PH * a.PW;
    const int nelem = a.G * CKS * plane_elems;
    int goff[EPT], loff[EPT];
    unsigned okbits = 0, oklast = 0, wrbits = 0;
    const int last_c0 = (a.nsteps_total - 1) * CKS;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NT;
        goff[i] = 0;
        loff[i] = 0;
        if (e < nelem) {
            const int pl = fdiv(e, a.m_plane), pos = e - pl * plane_elems;
            const int g = pl / CKS, c = pl - g * CKS;
            const int py = fdiv(pos, a.m_pw), px = pos - py * a.PW;
            const int gy = y_org + py, gx = x_org + px;
            const bool ok = ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W) & (n0 + g < a.N);
            if (ok) goff[i] = g * (int)a.in_n_stride + c * a.H * a.W + gy * a.W + gx;
            loff[i] = pl * a.PS + py * a.PWL + px;
            wrbits |= 1u << i;
            okbits |= (ok ? 1u : 0u) << i;
            oklast |= ((ok && last_c0 + c < a.Cin) ? 1u : 0u) << i;
        }
    }
    const int dummy_off = 2 * KD * BM + 2 * patch_floats + 4 * tid;  // floats from smem: this thread's 16-byte dummy slot
#pragma unroll
    for (int i = 1; i < EPT; ++i)
        if (!((wrbits >> i) & 1u)) goff[i] = goff[0];
    // ---- A loader: chunk q = tid + i*NT of the [KD][BM/4] tile; row r = tap*CKS + cl  <->  packed row tap*Cin + c0 + cl
    int aoff[APER];
#pragma unroll
    for (int i = 0; i < APER; ++i) {
        const int q = tid + i * NT;
        const int qc = q < A4 ? q : A4 - 1;
        const int r = qc / (BM / 4), c4 = qc - r * (BM / 4);
        aoff[i] = ((r / CKS) * a.Cin + (r % CKS)) * a.Mpad + m0 + c4 * 4;
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = M16 ? (lane & 15) : (lane & 31), lhi = M16 ? (lane >> 4) : (lane >> 5);  // row / pixel inside the MFMA block, k inside the group

    // ---- B fragment addressing: byte offset of this lane's pixel inside a patch buffer (+ odd-k plane)
    const int tile_pixels = a.G * a.TH * a.TW;
    int bbase[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int pj = (wn * TN + j) * PXW + l31;
        if (pj >= tile_pixels) pj = 0;  // padding lanes read a valid address; masked at the store
        const int g = fdiv(pj, a.m_thtw), rem = pj - g * (a.TH * a.TW);
        const int py = fdiv(rem, a.m_tw), px = rem - py * a.TW;
        bbase[j] = 4 * (g * CKS * a.PS + py * a.sh * a.PWL + px * a.sw + lhi * a.PS);
    }
    int so[NG];  // per k group: (first channel of the group)*PS + tap offset, bytes (wave uniform)
#pragma unroll
    for (int kk = 0; kk < NG; ++kk) so[kk] = 4 * (((KG * kk) % CKS) * a.PS + a.tapoff[cls][(KG * kk) / CKS]);

    AccT acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.0f;

    // two register sets: the global loads run TWO K-steps ahead of the MFMAs (the data of step s+1 is written
    // to LDS at the end of step s from the set that was loaded during step s-1), so a load has a full K-step
    // (~2300+ cycles) to return before anybody waits for it
    float pregA[EPT], pregB[EPT];
    floatx4 aregA[APER], aregB[APER];
    // Branch-free staging: slots past the end of the patch / A tile (the last i of a thread) load this thread's own element 0
    // again (a valid, thread-private address: one shared dummy address for every wave would be an L2 hot spot) and store it
    // to a private dummy slot behind the tiles, so the K loop carries no exec-mask branches.
    auto load_patch_one = [&](float (&preg)[EPT], int i, const float *__restrict__ base) { preg[i] = base[goff[i]]; };
    auto load_a_one = [&](floatx4 (&areg)[APER], int i, const float *__restrict__ base) {
        areg[i] = *reinterpret_cast<const floatx4 *>(base + aoff[i]);
    };
    auto store_patch_one = [&](const float (&preg)[EPT], int i, int buf, unsigned ok) {
        smem[((wrbits >> i) & 1u) ? 2 * KD * BM + buf * patch_floats + loff[i] : dummy_off] = ((ok >> i) & 1u) ? preg[i] : 0.0f;
    };
    auto store_a_one = [&](const floatx4 (&areg)[APER], int i, int buf) {
        *reinterpret_cast<floatx4 *>(smem + ((A4 % NT == 0 || tid + i * NT < A4) ? buf * (KD * BM) + (tid + i * NT) * 4 : dummy_off)) = areg[i];
    };
    auto store_tiles = [&](const float (&preg)[EPT], const floatx4 (&areg)[APER], int buf, unsigned ok) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) store_patch_one(preg, i, buf, ok);
#pragma unroll
        for (int i = 0; i < APER; ++i) store_a_one(areg, i, buf);
    };

    // Fragments of one K-step live in av / bv; the groups [0, G1) ("first half") of step s+1 are read from LDS while the second
    // half of step s is in the matrix pipe, the groups [G1, NG) of step s while its first half is -- so no MFMA ever waits for
    // an LDS read issued just before it, and the one barrier of the step sits BETWEEN the two halves:
    //   first half  : MFMAs [0,G1) | ds_read of groups [G1,NG) of this step | global loads of step s+2 | ds_write of step s+1
    //   barrier     : step s+1 is complete in the other LDS buffer (everybody finished reading it a step ago, see below)
    //   second half : MFMAs [G1,NG) | ds_read of groups [0,G1) of step s+1
    // Buffer safety: the buffer written during the first half of step s held step s-1; its last reads (second-half groups of
    // step s-1) were issued before the barrier of step s-1, which every wave has passed by now.
    constexpr int G1 = NG / 2 > 0 ? NG / 2 : 1, G2 = NG - G1;
    float av[NG][TM], bv[NG][TN];
    auto read_group = [&](int buf, int kk) {
        const char *Pb = reinterpret_cast<const char *>(Ps + buf * patch_floats);
        const float *A = As + buf * (KD * BM);
        const int k = KG * kk + lhi;
#pragma unroll
        for (int i = 0; i < TM; ++i) av[kk][i] = A[k * BM + (wm * TM + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[kk][j] = *reinterpret_cast<const float *>(Pb + bbase[j] + so[kk]);
    };
    auto mfma_group = [&](int g) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (M16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][i], bv[g][j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][i], bv[g][j], acc[i][j], 0, 0, 0);
            }
    };
    // LOADS: global loads of step `next` into (lpreg, lareg); STORE: (spreg, sareg) hold step s+1 and go to the other buffer
    // the loads walk through the K steps with two running pointers (no per-step index arithmetic in front of the MFMAs); a
    // run-ahead past the end of the slice stays on the last step (re-read, never used)
    const long pstride = (long)CKS * a.H * a.W, astride = (long)CKS * a.Mpad;
    const float *__restrict__ pnext = nullptr;
    const float *__restrict__ anext = nullptr;
    int lnext = 0, llast = 0;
    auto kstep = [&](int buf, float (&lpreg)[EPT], floatx4 (&lareg)[APER], const float (&spreg)[EPT],
                     const floatx4 (&sareg)[APER], unsigned sok, auto loads, auto store) {
        constexpr bool LOADS = decltype(loads)::value, STORE = decltype(store)::value;
        const float *__restrict__ pbase = pnext;
        const float *__restrict__ abase = anext;
        if (LOADS && lnext < llast) { pnext += pstride; anext += astride; ++lnext; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            mfma_group(g);
#pragma unroll
            for (int kk = G1 + g * G2 / G1; kk < G1 + (g + 1) * G2 / G1; ++kk) read_group(buf, kk);
            if (LOADS) {
#pragma unroll
                for (int i = g * EPT / G1; i < (g + 1) * EPT / G1; ++i) load_patch_one(lpreg, i, pbase);
#pragma unroll
                for (int i = g * APER / G1; i < (g + 1) * APER / G1; ++i) load_a_one(lareg, i, abase);
            }
            if (STORE) {
#pragma unroll
                for (int i = g * EPT / G1; i < (g + 1) * EPT / G1; ++i) store_patch_one(spreg, i, buf ^ 1, sok);
#pragma unroll
                for (int i = g * APER / G1; i < (g + 1) * APER / G1; ++i) store_a_one(sareg, i, buf ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STORE) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = G1; g < NG; ++g) {
            mfma_group(g);
            if (STORE) {
#pragma unroll
                for (int kk = (g - G1) * G1 / G2; kk < (g - G1 + 1) * G1 / G2; ++kk) read_group(buf ^ 1, kk);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int per_slice = (a.nsteps_total + a.ksplit - 1) / a.ksplit;
    const int s_begin = zs * per_slice;
    const int nsteps = min(a.nsteps_total, s_begin + per_slice) - s_begin;
    auto okmask_of = [&](int step) { return step == a.nsteps_total - 1 ? oklast : okbits; };
    auto phys = [&](int x) { return s_begin + min(x, nsteps - 1); };  // step of this K slice; a run-ahead past the end re-reads the last one
    if (nsteps > 0) {
        // prologue: step 0 -> LDS buffer 0; step 1 -> register set B (goes to LDS during step 0)
        const int p0s = phys(0);
        const float *__restrict__ pbase = in0 + (long)p0s * CKS * a.H * a.W;
        const float *__restrict__ abase = wp + (long)p0s * CKS * a.Mpad;
#pragma unroll
        for (int i = 0; i < EPT; ++i) load_patch_one(pregA, i, pbase);
#pragma unroll
        for (int i = 0; i < APER; ++i) load_a_one(aregA, i, abase);
        if (nsteps > 1) {  // a one-step slice has nothing to run ahead to
            const int s1 = phys(1);
            const float *__restrict__ pbase1 = in0 + (long)s1 * CKS * a.H * a.W;
            const float *__restrict__ abase1 = wp + (long)s1 * CKS * a.Mpad;
#pragma unroll
            for (int i = 0; i < EPT; ++i) load_patch_one(pregB, i, pbase1);
#pragma unroll
            for (int i = 0; i < APER; ++i) load_a_one(aregB, i, abase1);
        }
        store_tiles(pregA, aregA, 0, okmask_of(p0s));
    }
    __syncthreads();
    tl.mark(1);
    if (nsteps > 0) {
#pragma unroll
        for (int kk = 0; kk < G1; ++kk) read_group(0, kk);
    }
    // during step s (buffer s&1): MFMAs on step s, loads of step s+2 into the set that step s freed, the set holding step s+1
    // goes to LDS buffer (s+1)&1.  Pairs of steps keep the set indices static.
    int s = 0;
    if (nsteps > 0) {  // first step the loop loads: 2 (clamped to the last one)
        lnext = phys(2);
        llast = phys(nsteps - 1);
        pnext = in0 + (long)lnext * pstride;
        anext = wp + (long)lnext * astride;
    }
    for (; s + 2 < nsteps; s += 2) {
        kstep(0, pregA, aregA, pregB, aregB, okmask_of(phys(s + 1)), std::true_type{}, std::true_type{});
        kstep(1, pregB, aregB, pregA, aregA, okmask_of(phys(s + 2)), std::true_type{}, std::true_type{});
    }
    if (s + 1 < nsteps) {
        kstep(0, pregA, aregA, pregB, aregB, okmask_of(phys(s + 1)), std::false_type{}, std::true_type{});
        kstep(1, pregA, aregA, pregA, aregA, 0u, std::false_type{}, std::false_type{});
    } else if (nsteps > 0) {
        kstep(0, pregA, aregA, pregA, aregA, 0u, std::false_type{}, std::false_type{});
    }
    tl.mark(2);

    // ---- epilogue
    const int opy = cls >> 1, opx = cls & 1;
    const long plane = (long)a.Ho * a.Wo;
    const long Ptot = (long)a.N * a.Hp * a.Wp;
    // Fast path (plain convs whose tile rows and image rows are multiples of 4 pixels, all channels real): a 4x4
    // transpose inside every lane quad (two DPP butterfly stages) turns "lane = 1 pixel x 4 consecutive channels" into
    // "lane = 4 consecutive pixels x 1 channel", so each accumulator block leaves as ONE 16-byte store per lane
    // instead of four 4-byte stores (the store tail of a one-round grid is issue bound).
    const bool wide = a.ksplit == 1 && a.osx == 1 && a.osy == 1 && (a.TW & 3) == 0 && (a.Wp & 3) == 0 && (a.Cout & 3) == 0 && a.scale == nullptr;
    if (wide) {
        const int q = l31 >> 2, li = lane & 3;  // quad index inside the half wave, lane inside the quad
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pj = (wn * TN + j) * PXW + 4 * q;  // first of this lane's 4 pixels
            const bool pv = pj < tile_pixels;
            const int pjc = pv ? pj : 0;
            const int g = fdiv(pjc, a.m_thtw), rem = pjc - g * (a.TH * a.TW);
            const int py = fdiv(rem, a.m_tw), px = rem - py * a.TW;
            const int y = ty * a.TH + py, x = tx * a.TW + px, n = n0 + g;
            const bool ok = pv && y < a.Hp && x < a.Wp && n < a.N;
            float *__restrict__ ob = a.out + (long)n * a.out_n_stride + (long)y * a.Wo + x;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int rb = 0; rb < NR / 4; ++rb) {
                    float v0 = acc[i][j][4 * rb + 0], v1 = acc[i][j][4 * rb + 1], v2 = acc[i][j][4 * rb + 2], v3 = acc[i][j][4 * rb + 3];
                    // stage 1: exchange with the lane at distance 1 (quad_perm [1,0,3,2])
                    {
                        const bool odd = li & 1;
                        float s0 = odd ? v0 : v1, s1 = odd ? v2 : v3;
                        s0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
                        s1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
                        if (odd) { v0 = s0; v2 = s1; } else { v1 = s0; v3 = s1; }
                    }
                    // stage 2: exchange with the lane at distance 2 (quad_perm [2,3,0,1])
                    {
                        const bool hi = li & 2;
                        float s0 = hi ? v0 : v2, s1 = hi ? v1 : v3;
                        s0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, true));
                        s1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, true));
                        if (hi) { v0 = s0; v1 = s1; } else { v2 = s0; v3 = s1; }
                    }
                    // now this lane holds channel (row block + li) for pixels 4q .. 4q+3
                    const int co = m0 + (wm * TM + i) * 32 + li + 8 * rb + 4 * lhi;
                    if (ok && co < a.Cout) {
                        const float b = a.bias[co];
                        floatx4 v = {v0 + b, v1 + b, v2 + b, v3 + b};
                        if (a.act) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : 0.1f * v[e];
                        }
                        *reinterpret_cast<floatx4 *>(ob + (long)co * plane) = v;
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pj = (wn * TN + j) * PXW + l31;
        if (pj >= tile_pixels) continue;
        const int g = fdiv(pj, a.m_thtw), rem = pj - g * (a.TH * a.TW);
        const int py = fdiv(rem, a.m_tw), px = rem - py * a.TW;
        const int y = ty * a.TH + py, x = tx * a.TW + px, n = n0 + g;
        if (y >= a.Hp || x >= a.Wp || n >= a.N) continue;
        if (a.ksplit > 1) {
            const long p = ((long)n * a.Hp + y) * a.Wp + x;
            float *__restrict__ ws = a.ws + ((long)blockIdx.z * a.Mpad) * Ptot + p;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int co = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    ws[(long)co * Ptot] = acc[i][j][r];
                }
        } else {
            float *__restrict__ ob = a.out + (long)n * a.out_n_stride + (long)(y * a.osy + opy) * a.Wo + (x * a.osx + opx);
            const float sc = a.scale ? a.scale[n] : 1.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int co = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (co < a.Cout) {
                        float v = acc[i][j][r] + a.bias[co];
                        if (a.act) v = v >= 0.0f ? v : 0.1f * v;
                        if (co == 0) v *= sc;
                        ob[(long)co * plane] = v;
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// deconv4_kernel: all FOUR sub-pixel classes of a 4x4 stride-2 transposed conv in one workgroup (blocks_original.py:64-75, :97-110).
//
// The per-class launch of conv_patch_kernel (blockIdx.z = class) reads the input once per class and lets every class write every
// second pixel of an output row: on the refinement net's big maps that is 11x the algorithmic HBM reads (4 classes x halo, no L2
// reuse between z planes, read-for-ownership of half-written lines) and 2x the writes (rocprofv3 FETCH_SIZE / WRITE_SIZE).  Here a
// workgroup stages the union patch of its input pixel tile ((TH+2) x (TW+2), taps dy,dx in {-1,0,+1}) ONCE per K-step, runs the
// 4 x (4 taps x CKS channels) MFMA groups of all classes on it (4x the MFMAs per staged float), and writes full output rows: a
// lane owns input pixel (y,x) and stores the pixel PAIRS (2y+py, 2x..2x+1) as 8 bytes, 32 lanes = 256 contiguous bytes.
// Wave tile 32 channels x 32 pixels per class (TM = TN = 1), accumulators 4 classes x 16 registers.
// Split-K partials use the layout of the per-class path ([cls][slice][Mpad][P]), so conv_splitk_reduce_kernel finishes them.
template <int BM, int WM, int WN, int EPT>
__global__ __launch_bounds__(64 * WM * WN) void deconv4_kernel(PatchArgs a)
{
    constexpr int NT = 64 * WM * WN;
    constexpr int CKS = 4, KD = 16, NG = 8;      // per class: 4 taps x 4 channels
    constexpr int A4 = 4 * KD * BM / 4;          // float4 chunks of the A tiles of the four classes
    constexpr int APER = (A4 + NT - 1) / NT;
    static_assert(BM == WM * 32, "bad tile");

    TlScope tl(a.tl);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                            // [2][4][KD][BM]
    float *Ps = smem + 2 * 4 * KD * BM;          // [2][patch_floats]
    const int patch_floats = a.G * CKS * a.PS;

    const int tid = threadIdx.x;
    const int zs = blockIdx.z;
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
    const int m0 = by * BM;
    const int tyg = fdiv((int)bx, a.m_tilesx);
    const int tx = (int)bx - tyg * a.tiles_x;
    const int tgrp = fdiv(tyg, a.m_tilesy);
    const int ty = tyg - tgrp * a.tiles_y;
    const int n0 = tgrp * a.G;
    const int y_org = ty * a.TH - 1, x_org = tx * a.TW - 1;
    const float *__restrict__ in0 = a.in + (long)n0 * a.in_n_stride;

    const int plane_elems = a.PH * a.PW;
    const int nelem = a.G * CKS * plane_elems;
    int goff[EPT], loff[EPT];
    unsigned okbits = 0, oklast = 0, wrbits = 0;
    const int last_c0 = (a.nsteps_total - 1) * CKS;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NT;
        goff[i] = 0;
        loff[i] = 0;
        if (e < nelem) {
            const int pl = fdiv(e, a.m_plane), pos = e - pl * plane_elems;
            const int g = pl / CKS, c = pl - g * CKS;
            const int py = fdiv(pos, a.m_pw), px = pos - py * a.PW;
            const int gy = y_org + py, gx = x_org + px;
            const bool ok = ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W) & (n0 + g < a.N);
            if (ok) goff[i] = g * (int)a.in_n_stride + c * a.H * a.W + gy * a.W + gx;
            loff[i] = pl * a.PS + py * a.PWL + px;
            wrbits |= 1u << i;
            okbits |= (ok ? 1u : 0u) << i;
            oklast |= ((ok && last_c0 + c < a.Cin) ? 1u : 0u) << i;
        }
    }
    const int dummy_off = 2 * 4 * KD * BM + 2 * patch_floats + 4 * tid;  // floats from smem: this thread's 16-byte dummy slot
#pragma unroll
    for (int i = 1; i < EPT; ++i)
        if (!((wrbits >> i) & 1u)) goff[i] = goff[0];
    // A loader: chunk q of the [cls][KD][BM/4] tiles; row r = tap*CKS + cl  <->  packed row tap*Cin + c0 + cl of class cls
    long aoff[APER];
#pragma unroll
    for (int i = 0; i < APER; ++i) {
        const int q = min(tid + i * NT, A4 - 1);
        const int cls = q / (KD * BM / 4), qq = q - cls * (KD * BM / 4);
        const int r = qq / (BM / 4), c4 = qq - r * (BM / 4);
        aoff[i] = (long)cls * a.cls_w_stride + (long)((r / CKS) * a.Cin + (r % CKS)) * a.Mpad + m0 + c4 * 4;
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tile_pixels = a.G * a.TH * a.TW;
    int pj = wn * 32 + l31;
    const bool pvalid = pj < tile_pixels;
    if (!pvalid) pj = 0;
    const int pg = fdiv(pj, a.m_thtw), prem = pj - pg * (a.TH * a.TW);
    const int ppy = fdiv(prem, a.m_tw), ppx = prem - ppy * a.TW;
    const int bbase = 4 * (pg * CKS * a.PS + ppy * a.PWL + ppx + lhi * a.PS);

    floatx16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

    float pregA[EPT], pregB[EPT];
    floatx4 aregA[APER], aregB[APER];
    // branch-free staging as in conv_patch_kernel: slots past the end re-load the thread's element 0 and land in a private dummy slot
    auto load_patch_one = [&](float (&preg)[EPT], int i, const float *__restrict__ base) { preg[i] = base[goff[i]]; };
    auto load_a_one = [&](floatx4 (&areg)[APER], int i, const float *__restrict__ base) {
        areg[i] = *reinterpret_cast<const floatx4 *>(base + aoff[i]);
    };
    auto store_tiles = [&](const float (&preg)[EPT], const floatx4 (&areg)[APER], int buf, unsigned ok) {
#pragma unroll
        for (int i = 0; i < EPT; ++i)
            smem[((wrbits >> i) & 1u) ? 2 * 4 * KD * BM + buf * patch_floats + loff[i] : dummy_off] = ((ok >> i) & 1u) ? preg[i] : 0.0f;
#pragma unroll
        for (int i = 0; i < APER; ++i)
            *reinterpret_cast<floatx4 *>(smem + ((A4 % NT == 0 || tid + i * NT < A4) ? buf * (4 * KD * BM) + (tid + i * NT) * 4 : dummy_off)) = areg[i];
    };
    auto kstep = [&](int buf, int next, float (&preg)[EPT], floatx4 (&areg)[APER], auto prefetch) {
        constexpr bool PREFETCH = decltype(prefetch)::value;
        const char *Pb = reinterpret_cast<const char *>(Ps + buf * patch_floats) + bbase;
        const float *A = As + buf * (4 * KD * BM) + wm * 32 + l31;
        const float *__restrict__ pbase = in0 + (long)next * CKS * a.H * a.W;
        const float *__restrict__ abase = a.wp + (long)next * CKS * a.Mpad;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float av[NG], bv[NG];
#pragma unroll
            for (int kk = 0; kk < NG; ++kk) {
                const int k = 2 * kk + lhi;
                av[kk] = A[(c * KD + k) * BM];
                // k pair kk = tap kk/2, channels 2*(kk&1) + lhi (lhi is part of bbase)
                bv[kk] = *reinterpret_cast<const float *>(Pb + 4 * ((2 * (kk & 1)) * a.PS + a.tapoff[c][kk >> 1]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g], bv[g], acc[c], 0, 0, 0);
                if (PREFETCH && c < 2) {  // the loads of step s+2 ride behind the MFMAs of the first two classes
                    constexpr int SL = 2 * NG;
                    const int slot = c * NG + g;
#pragma unroll
                    for (int i = slot * EPT / SL; i < (slot + 1) * EPT / SL; ++i) load_patch_one(preg, i, pbase);
#pragma unroll
                    for (int i = slot * APER / SL; i < (slot + 1) * APER / SL; ++i) load_a_one(areg, i, abase);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int per_slice = (a.nsteps_total + a.ksplit - 1) / a.ksplit;
    const int s_begin = zs * per_slice;
    const int nsteps = min(a.nsteps_total, s_begin + per_slice) - s_begin;
    auto okmask_of = [&](int step) { return step == a.nsteps_total - 1 ? oklast : okbits; };
    auto phys = [&](int x) { return s_begin + min(x, nsteps - 1); };
    if (nsteps > 0) {
        const int p0s = phys(0);
        const float *__restrict__ pbase = in0 + (long)p0s * CKS * a.H * a.W;
        const float *__restrict__ abase = a.wp + (long)p0s * CKS * a.Mpad;
#pragma unroll
        for (int i = 0; i < EPT; ++i) load_patch_one(pregA, i, pbase);
#pragma unroll
        for (int i = 0; i < APER; ++i) load_a_one(aregA, i, abase);
        if (nsteps > 1) {
            const int s1 = phys(1);
            const float *__restrict__ pbase1 = in0 + (long)s1 * CKS * a.H * a.W;
            const float *__restrict__ abase1 = a.wp + (long)s1 * CKS * a.Mpad;
#pragma unroll
            for (int i = 0; i < EPT; ++i) load_patch_one(pregB, i, pbase1);
#pragma unroll
            for (int i = 0; i < APER; ++i) load_a_one(aregB, i, abase1);
        }
        store_tiles(pregA, aregA, 0, okmask_of(p0s));
    }
    __syncthreads();
    tl.mark(1);
    int s = 0;
    for (; s + 2 < nsteps; s += 2) {
        kstep(0, phys(s + 2), pregA, aregA, std::true_type{});
        store_tiles(pregB, aregB, 1, okmask_of(phys(s + 1)));
        __syncthreads();
        kstep(1, phys(s + 3), pregB, aregB, std::true_type{});
        store_tiles(pregA, aregA, 0, okmask_of(phys(s + 2)));
        __syncthreads();
    }
    if (s + 1 < nsteps) {
        kstep(0, 0, pregA, aregA, std::false_type{});
        store_tiles(pregB, aregB, 1, okmask_of(phys(s + 1)));
        __syncthreads();
        ++s;
    }
    if (nsteps > 0) kstep(s & 1, 0, pregA, aregA, std::false_type{});
    tl.mark(2);

    // ---- epilogue: lane = input pixel (y, x) of the tile
    const int y = ty * a.TH + ppy, x = tx * a.TW + ppx, n = n0 + pg;
    if (!pvalid || y >= a.Hp || x >= a.Wp || n >= a.N) return;
    if (a.ksplit > 1) {
        const long Ptot = (long)a.N * a.Hp * a.Wp;
        const long p = ((long)n * a.Hp + y) * a.Wp + x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float *__restrict__ ws = a.ws + (((long)c * a.ksplit + zs) * a.Mpad) * Ptot + p;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                ws[(long)co * Ptot] = acc[c][r];
            }
        }
        return;
    }
    const long plane = (long)a.Ho * a.Wo;
    float *__restrict__ ob = a.out + (long)n * a.out_n_stride + (long)(2 * y) * a.Wo + 2 * x;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (co < a.Cout) {
            const float b = a.bias[co];
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                float v0 = acc[2 * py][r] + b, v1 = acc[2 * py + 1][r] + b;
                if (a.act) {
                    v0 = v0 >= 0.0f ? v0 : 0.1f * v0;
                    v1 = v1 >= 0.0f ? v1 : 0.1f * v1;
                }
                float2 v = {v0, v1};
                *reinterpret_cast<float2 *>(ob + (long)co * plane + (long)py * a.Wo) = v;
            }
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------
struct PatchTile { int bm, bn, threads; };
static const PatchTile kPatchTiles[PTILE_COUNT] = {{128, 128, 256}, {64, 128, 256}, {32, 128, 256}, {64, 64, 256}, {128, 64, 256}, {32, 64, 128}, {16, 128, 256}, {32, 128, 256}, {64, 64, 256}};

bool patch_tile_is_dc4(int tile) { return tile == PTILE_DC4_32x128 || tile == PTILE_DC4_64x64; }

int patch_cks(int ntaps, int tile)
{
    if (patch_tile_is_dc4(tile)) return ntaps == 4 ? 4 : 0;
    if (kPatchTiles[tile].bm == 16) return ntaps == 9 ? 4 : 0;  // 16-row MFMA tile: 3x3 heads only (K groups of 4 channels)
    switch (ntaps) {
        case 3: return 8;
        case 4: return 4;
        case 5: return 4;
        case 7: return 2;
        case 9: return 2;
        default: return 0;
    }
}

int patch_tile_bm(int tile) { return kPatchTiles[tile].bm; }
int patch_tile_bn(int tile) { return kPatchTiles[tile].bn; }
int patch_tile_threads(int tile) { return kPatchTiles[tile].threads; }
int patch_tile_mtiles(int tile, int Cout, int Mpad) { return kPatchTiles[tile].bm == 16 ? (Cout + 15) / 16 : Mpad / kPatchTiles[tile].bm; }

size_t patch_lds_bytes(int tile, int ntaps, int G, int PS)
{
    const int cks = patch_cks(ntaps, tile);
    const size_t a_tiles = patch_tile_is_dc4(tile) ? 4 : 1;  // the fused transposed conv stages the weights of its four classes
    // + one 16-byte dummy slot per thread (branch-free staging)
    return sizeof(float) * (2ul * a_tiles * ntaps * cks * kPatchTiles[tile].bm + 2ul * G * cks * PS + 4ul * kPatchTiles[tile].threads);
}

template <int BM, int WM, int WN, int TM, int TN, int EPT>
static void launch_patch_taps(const PatchArgs &a, int ntaps, dim3 grid, size_t lds, hipStream_t s)
{
    constexpr int NT = 64 * WM * WN;
    switch (ntaps) {
        case 3: hipLaunchKernelGGL((conv_patch_kernel<BM, WM, WN, TM, TN, 3, 8, EPT>), grid, dim3(NT), lds, s, a); break;
        case 4: hipLaunchKernelGGL((conv_patch_kernel<BM, WM, WN, TM, TN, 4, 4, EPT>), grid, dim3(NT), lds, s, a); break;
        case 5: hipLaunchKernelGGL((conv_patch_kernel<BM, WM, WN, TM, TN, 5, 4, EPT>), grid, dim3(NT), lds, s, a); break;
        case 7: hipLaunchKernelGGL((conv_patch_kernel<BM, WM, WN, TM, TN, 7, 2, EPT>), grid, dim3(NT), lds, s, a); break;
        case 9: hipLaunchKernelGGL((conv_patch_kernel<BM, WM, WN, TM, TN, 9, 2, EPT>), grid, dim3(NT), lds, s, a); break;
        default: break;
    }
}

template <int BM, int WM, int WN>
static void launch_dc4(const PatchArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    const long elems = (long)a.G * 4 * a.PH * a.PW;
    const int per_thread = (int)((elems + 64 * WM * WN - 1) / (64 * WM * WN));
    if (per_thread <= 2) hipLaunchKernelGGL((deconv4_kernel<BM, WM, WN, 2>), grid, dim3(64 * WM * WN), lds, s, a);
    else if (per_thread <= 4) hipLaunchKernelGGL((deconv4_kernel<BM, WM, WN, 4>), grid, dim3(64 * WM * WN), lds, s, a);
    else hipLaunchKernelGGL((deconv4_kernel<BM, WM, WN, PATCH_EPT>), grid, dim3(64 * WM * WN), lds, s, a);
}

template <int EPT>
static void launch_patch16(const PatchArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL((conv_patch_kernel<16, 1, 4, 1, 2, 9, 4, EPT>), grid, dim3(256), lds, s, a);
}

template <int BM, int WM, int WN, int TM, int TN>
static void launch_patch_ept(const PatchArgs &a, int ntaps, int cks, dim3 grid, size_t lds, hipStream_t s)
{
    const long elems = (long)a.G * cks * a.PH * a.PW;
    const int per_thread = (int)((elems + 64 * WM * WN - 1) / (64 * WM * WN));
    if (per_thread <= 2) launch_patch_taps<BM, WM, WN, TM, TN, 2>(a, ntaps, grid, lds, s);
    else if (per_thread <= 4) launch_patch_taps<BM, WM, WN, TM, TN, 4>(a, ntaps, grid, lds, s);
    else launch_patch_taps<BM, WM, WN, TM, TN, PATCH_EPT>(a, ntaps, grid, lds, s);
}

void launch_conv_patch(const PatchArgs &a, int tile, int ntaps, int nclasses, hipStream_t stream)
{
    const int groups = (a.N + a.G - 1) / a.G;
    dim3 grid((unsigned)(groups * a.tiles_y * a.tiles_x), (unsigned)patch_tile_mtiles(tile, a.Cout, a.Mpad), (unsigned)(nclasses * a.ksplit));
    const size_t lds = patch_lds_bytes(tile, ntaps, a.G, a.PS);
    const int cks = patch_cks(ntaps, tile);
    if (patch_tile_is_dc4(tile)) {
        grid.z = (unsigned)a.ksplit;  // the classes are a loop inside the workgroup
        if (tile == PTILE_DC4_32x128) launch_dc4<32, 1, 4>(a, grid, lds, stream);
        else launch_dc4<64, 2, 2>(a, grid, lds, stream);
        return;
    }
    switch (tile) {
        case PTILE_128x128: launch_patch_ept<128, 2, 2, 2, 2>(a, ntaps, cks, grid, lds, stream); break;
        case PTILE_64x128:  launch_patch_ept<64, 2, 2, 1, 2>(a, ntaps, cks, grid, lds, stream); break;
        case PTILE_32x128:  launch_patch_ept<32, 1, 4, 1, 1>(a, ntaps, cks, grid, lds, stream); break;
        case PTILE_128x64:  launch_patch_ept<128, 2, 2, 2, 1>(a, ntaps, cks, grid, lds, stream); break;
        case PTILE_32x64:   launch_patch_ept<32, 1, 2, 1, 1>(a, ntaps, cks, grid, lds, stream); break;
        case PTILE_16x128: {
            const long elems = (long)a.G * cks * a.PH * a.PW;
            const int per_thread = (int)((elems + 255) / 256);
            if (per_thread <= 2) launch_patch16<2>(a, grid, lds, stream);
            else if (per_thread <= 4) launch_patch16<4>(a, grid, lds, stream);
            else launch_patch16<PATCH_EPT>(a, grid, lds, stream);
            break;
        }
        default:            launch_patch_ept<64, 2, 2, 1, 1>(a, ntaps, cks, grid, lds, stream); break;
    }
}

}  // namespace demon
