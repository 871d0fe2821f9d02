// conv_mfma.hip -- implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
// One kernel family serves every contraction of the DeMoN nets:
//   * k x 1 / 1 x k strided convs of helpers.py:105-153 (convrelu2_caffe_padding)
//   * 3 x 3 convs of helpers.py:70-102
//   * 4 x 4 stride-2 transposed convs of blocks_original.py:64-75, :97-110, as four 2x2 sub-pixel
//     convolutions (blockIdx.z = output parity class)
//   * the dense layers of blocks_original.py:390-410 (H = W = 1)
//
// GEMM view:  D[co][pix] = sum_k Wp[k][co] * X[k][pix]
//   A = packed weights, M = Cout (rows of the 32x32 MFMA tile)
//   B = im2col of the NCHW input gathered on the fly through a per-layer K table, N = pixels
//   so that the 32 lanes of an accumulator register cover 32 consecutive pixels of one output
//   channel: NCHW stores are 128-byte coalesced with no transposition.
// v_mfma_f32_32x32x2_f32: lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; the result
// register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].  fp32 in, fp32 out,
// bit-identical to an fmaf chain in k order (cdna_hip_programming.md section 3).
//
// Pipeline: global -> registers (prefetch of K-step s+1) overlapped with the MFMAs of K-step s out
// of a double-buffered LDS tile, one barrier per K-step.  A 64-cycle MFMA needs one 4-byte LDS read
// per operand per lane, so LDS bandwidth is never the limit; the tile shapes below only trade
// operand reuse (global -> LDS traffic) against the number of workgroups for the small feature maps.
#include <type_traits>

#include <stdio.h>
#include <stdlib.h>

#include "internal.h"

namespace demon {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvArgs a)
{
    constexpr int BK = 16;
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int BROWS = NT / BN;             // k rows covered by one pass of the B staging
    constexpr int BPER = BK / BROWS;           // B elements per thread per K-step
    constexpr int ACH = BM / 4;                // float4 chunks per A row
    constexpr int AROWS = NT / ACH;            // rows covered by one pass of the A staging
    constexpr int APER = (BK + AROWS - 1) / AROWS;
    static_assert(NT % BN == 0 && BK % BROWS == 0, "bad B staging shape");

    TlScope tl(a.tl);
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int cls = blockIdx.z / a.ksplit;  // output parity class (transposed conv) ...
    const int zs = blockIdx.z - cls * a.ksplit;  // ... and K slice (split-K for the small feature maps)
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
    const int m0 = by * BM;
    const long p0 = (long)bx * BN;
    const long P = (long)a.N * a.Hp * a.Wp;
    const float *__restrict__ wp = a.wp + (long)cls * a.cls_w_stride;
    const KEntry *__restrict__ ktab = a.ktab + (long)cls * a.Kpad;

    // ---- B staging: this thread gathers pixel column bj for k rows bg, bg+BROWS, ...
    const int bj = tid % BN;
    int bg = tid / BN;
    if (BN >= 64) bg = __builtin_amdgcn_readfirstlane(bg);  // wave-uniform: K table reads go scalar
    int iy0, ix0;
    const float *__restrict__ inb;
    {
        const long p = p0 + bj;
        if (p < P) {
            const int x = (int)(p % a.Wp);
            const long t = p / a.Wp;
            const int y = (int)(t % a.Hp);
            const int n = (int)(t / a.Hp);
            iy0 = y * a.sy;
            ix0 = x * a.sx;
            inb = a.in + (long)n * a.in_n_stride + (long)iy0 * a.W + ix0;
        } else {
            iy0 = -(1 << 20);  // every bounds test fails -> zeros
            ix0 = 0;
            inb = a.in;
        }
    }
    // ---- A staging
    const int ac4 = tid % ACH;
    const int arow = tid / ACH;

    float breg[BPER];
    floatx4 areg[APER];
    KEntry kentA[BPER], kentB[BPER];  // K-table entries, two K-steps in flight (even / odd step)
    unsigned okmask = 0;              // bit i: B element i of the prefetched K-step is inside the image

    // The prefetch of K-step s+1 is cut into per-element pieces that are issued in the shadow of the
    // MFMA groups of K-step s (a 64-cycle fp32 MFMA hides ~10 other instructions of the same wave):
    //   load_ktab   one scalar (vector for 32-pixel tiles) burst with this thread's BPER table entries.
    //               Every K-step touches a fresh 128-byte piece of the table, i.e. a scalar-cache miss of
    //               ~1000 cycles that the lock-stepped waves of a SIMD would all sit out together, so the
    //               table runs TWO steps ahead of the MFMAs (one step ahead of the gathers that use it).
    //   gather_one  branch-free gather: out-of-image taps read the (always valid) anchor pixel and are
    //               zeroed when written to LDS, so no load sits behind an exec-mask branch
    auto load_ktab = [&](KEntry (&kent)[BPER], int step) {
        step = min(step, a.Kpad / BK - 1);  // the run-ahead may point past the table: re-read the last step
#pragma unroll
        for (int i = 0; i < BPER; ++i) kent[i] = ktab[step * BK + bg * BPER + i];
    };
    auto gather_one = [&](const KEntry (&kent)[BPER], int i) {
        const int dy = kent[i].dydx >> 16;
        const int dx = (int)(short)(kent[i].dydx & 0xffff);
        const bool ok = ((unsigned)(iy0 + dy) < (unsigned)a.H) & ((unsigned)(ix0 + dx) < (unsigned)a.W);
        okmask |= (ok ? 1u : 0u) << i;
        breg[i] = inb[ok ? kent[i].delta : 0];
    };
    auto load_a_one = [&](int i, int step) {
        const int row = arow + i * AROWS;
        if (AROWS * APER == BK || row < BK)
            areg[i] = *reinterpret_cast<const floatx4 *>(wp + (long)(step * BK + row) * a.Mpad + m0 + ac4 * 4);
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BPER; ++i) Bs[buf][bg * BPER + i][bj] = ((okmask >> i) & 1u) ? breg[i] : 0.0f;
#pragma unroll
        for (int i = 0; i < APER; ++i) {
            const int row = arow + i * AROWS;
            if (AROWS * APER == BK || row < BK) *reinterpret_cast<floatx4 *>(&As[buf][row][ac4 * 4]) = areg[i];
        }
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // One K-step on LDS buffer `buf`: the fragments of the whole step, then 8 MFMA groups.  With PREFETCH the
    // global loads of step `next` (table entries in kent_use) and the table read of step next+1 (into
    // kent_load) are pinned between the groups; sched_barrier keeps that order.
    constexpr int NG = BK / 2;
    auto kstep = [&](int buf, int next, const KEntry (&kent_use)[BPER], KEntry (&kent_load)[BPER], auto prefetch) {
        constexpr bool PREFETCH = decltype(prefetch)::value;
        float av[NG][TM], bv[NG][TN];
#pragma unroll
        for (int kk = 0; kk < NG; ++kk) {
            const int k = 2 * kk + lhi;
#pragma unroll
            for (int i = 0; i < TM; ++i) av[kk][i] = As[buf][k][(wm * TM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[kk][j] = Bs[buf][k][(wn * TN + j) * 32 + l31];
        }
        if (PREFETCH) {
            okmask = 0;
            // all LDS reads back before the first MFMA (~150 cycles, once per step): the scalar table read
            // issued below shares lgkmcnt with them and returns out of order, so any later LDS wait would
            // turn into lgkmcnt(0) and sit out the table's cache miss
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][i], bv[g][j], acc[i][j], 0, 0, 0);
            if (PREFETCH) {
                // the gathers go behind the FIRST half of the groups so that the last one still has ~1000
                // cycles of MFMA work to land before store_tiles waits for it (lock-stepped waves do not
                // cover each other's waits)
                constexpr int GH = NG / 2;
                if (g < GH) {
#pragma unroll
                    for (int i = g * BPER / GH; i < (g + 1) * BPER / GH; ++i) gather_one(kent_use, i);
                }
                if (g < APER) load_a_one(g, next);
                // table entries of step next+1 into the register set that step next-1 used.  Issued here,
                // behind this step's LDS reads, so that no lgkmcnt(0) wait of this step covers the miss.
                if (g == GH) load_ktab(kent_load, next + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int total_steps = a.Kpad / BK;
    const int per_slice = (total_steps + a.ksplit - 1) / a.ksplit;
    const int s_begin = zs * per_slice;
    const int nsteps = min(total_steps, s_begin + per_slice) - s_begin;  // may be <= 0 for a trailing slice
    if (nsteps > 0) {
        load_ktab(kentA, s_begin);
        load_ktab(kentB, s_begin + 1);
#pragma unroll
        for (int i = 0; i < BPER; ++i) gather_one(kentA, i);
#pragma unroll
        for (int i = 0; i < APER; ++i) load_a_one(i, s_begin);
        store_tiles(0);
    }
    __syncthreads();
    tl.mark(1);
    // steps in pairs so that the table double buffer is indexed statically: during an even step s the gathers
    // of step s+1 read kentB while kentA is refilled for step s+2, and vice versa during odd steps
    int s = 0;
    for (; s + 2 < nsteps; s += 2) {
        kstep(0, s_begin + s + 1, kentB, kentA, std::true_type{});
        store_tiles(1);
        __syncthreads();
        kstep(1, s_begin + s + 2, kentA, kentB, std::true_type{});
        store_tiles(0);
        __syncthreads();
    }
    if (s + 1 < nsteps) {
        kstep(0, s_begin + s + 1, kentB, kentA, std::true_type{});
        store_tiles(1);
        __syncthreads();
        ++s;
    }
    if (nsteps > 0) kstep(s & 1, 0, kentA, kentB, std::false_type{});
    tl.mark(2);

    if (a.ksplit > 1) {  // raw partial sums to the workspace [cls][slice][Mpad][P]; conv_splitk_reduce finishes
        float *__restrict__ ws = a.ws + ((long)blockIdx.z * a.Mpad) * P;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long p = p0 + (wn * TN + j) * 32 + l31;
            if (p >= P) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    ws[(long)co * P + p] = acc[i][j][r];
                }
        }
        return;
    }
    // ---- epilogue: bias, leaky relu, optional per-sample scale of channel 0, coalesced NCHW store
    const int pyc = cls >> 1, pxc = cls & 1;  // cls = 0 for plain convs
    const long plane = a.out_plane;
    // Fast path (plain convs, image rows a multiple of 4 pixels, all channels real): 4x4 transpose inside every lane quad
    // (two DPP butterfly stages) so that each lane stores 4 consecutive pixels of one channel with one 16-byte store
    if (a.osx == 1 && a.osy == 1 && (a.Wp & 3) == 0 && (a.Cout & 3) == 0 && a.scale == nullptr) {
        const int q = l31 >> 2, li = lane & 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long p = p0 + (wn * TN + j) * 32 + 4 * q;
            const bool ok = p < P;
            const long pc = ok ? p : 0;
            const int x = (int)(pc % a.Wp);
            const long t = pc / a.Wp;
            const int y = (int)(t % a.Hp);
            const int n = (int)(t / a.Hp);
            float *__restrict__ ob = a.out + (long)n * a.out_n_stride + (long)y * a.Wo + x;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    float v0 = acc[i][j][4 * rb + 0], v1 = acc[i][j][4 * rb + 1], v2 = acc[i][j][4 * rb + 2], v3 = acc[i][j][4 * rb + 3];
                    {
                        const bool odd = li & 1;  // exchange with the lane at distance 1 (quad_perm [1,0,3,2])
                        float s0 = odd ? v0 : v1, s1 = odd ? v2 : v3;
                        s0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
                        s1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
                        if (odd) { v0 = s0; v2 = s1; } else { v1 = s0; v3 = s1; }
                    }
                    {
                        const bool hi = li & 2;  // exchange with the lane at distance 2 (quad_perm [2,3,0,1])
                        float s0 = hi ? v0 : v2, s1 = hi ? v1 : v3;
                        s0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, true));
                        s1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, true));
                        if (hi) { v0 = s0; v1 = s1; } else { v2 = s0; v3 = s1; }
                    }
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
        const long p = p0 + (wn * TN + j) * 32 + l31;
        if (p >= P) continue;
        const int x = (int)(p % a.Wp);
        const long t = p / a.Wp;
        const int y = (int)(t % a.Hp);
        const int n = (int)(t / a.Hp);
        float *__restrict__ ob = a.out + (long)n * a.out_n_stride + (long)(y * a.osy + pyc) * a.Wo + (x * a.osx + pxc);
        const float sc = a.scale ? a.scale[n] : 1.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
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

// sums the split-K slices, then the same epilogue as the fused path.  grid: (ceil(P/256), Cout, ncls)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvArgs a)
{
    const long P = (long)a.N * a.Hp * a.Wp;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int co = blockIdx.y, cls = blockIdx.z;
    const float *__restrict__ ws = a.ws + (((long)cls * a.ksplit) * a.Mpad + co) * P + p;
    // four slices per round: a reduce launch is a chain of memory latencies (one per dependent load), not a stream; the order of the
    // additions stays slice 0, 1, 2 ... (what the in-launch form and the tests expect)
    const long zs = (long)a.Mpad * P;
    float v = 0.0f;
    int z = 0;
    for (; z + 4 <= a.ksplit; z += 4) {
        const float s0 = ws[z * zs], s1 = ws[(z + 1) * zs], s2 = ws[(z + 2) * zs], s3 = ws[(z + 3) * zs];
        v = (((v + s0) + s1) + s2) + s3;
    }
    for (; z < a.ksplit; ++z) v += ws[z * zs];
    v += a.bias[co];
    if (a.act) v = v >= 0.0f ? v : 0.1f * v;
    const int x = (int)(p % a.Wp);
    const long t = p / a.Wp;
    const int y = (int)(t % a.Hp);
    const int n = (int)(t / a.Hp);
    if (co == 0 && a.scale) v *= a.scale[n];
    a.out[(long)n * a.out_n_stride + (long)co * a.out_plane + (long)(y * a.osy + (cls >> 1)) * a.Wo + (x * a.osx + (cls & 1))] = v;
}

// the same for plain convs whose rows are a multiple of 4 pixels: a thread sums four consecutive pixels (16-byte loads and
// stores, a quarter of the workgroups); per-element summation order unchanged.  grid: (ceil(P/1024), Cout)
__global__ __launch_bounds__(256) void conv_splitk_reduce4_kernel(ConvArgs a)
{
    const long P = (long)a.N * a.Hp * a.Wp;
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= P) return;
    const int co = blockIdx.y;
    const float *__restrict__ ws = a.ws + (long)co * P + p;
    const long zs = (long)a.Mpad * P;
    floatx4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    int z = 0;
    for (; z + 4 <= a.ksplit; z += 4) {   // (four loads in flight, additions in slice order)
        const floatx4 s0 = *reinterpret_cast<const floatx4 *>(ws + z * zs), s1 = *reinterpret_cast<const floatx4 *>(ws + (z + 1) * zs);
        const floatx4 s2 = *reinterpret_cast<const floatx4 *>(ws + (z + 2) * zs), s3 = *reinterpret_cast<const floatx4 *>(ws + (z + 3) * zs);
        v = (((v + s0) + s1) + s2) + s3;
    }
    for (; z < a.ksplit; ++z) v += *reinterpret_cast<const floatx4 *>(ws + z * zs);
    const float b = a.bias[co];
    const int x = (int)(p % a.Wp);
    const long t = p / a.Wp;
    const int y = (int)(t % a.Hp);
    const int n = (int)(t / a.Hp);
    const float sc = (co == 0 && a.scale) ? a.scale[n] : 1.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r = v[e] + b;
        if (a.act) r = r >= 0.0f ? r : 0.1f * r;
        if (co == 0 && a.scale) r *= sc;
        v[e] = r;
    }
    *reinterpret_cast<floatx4 *>(a.out + (long)n * a.out_n_stride + (long)co * a.out_plane + (long)y * a.Wo + x) = v;
}

struct TileInfo { int bm, bn, threads; float eff; };
static const TileInfo kTiles[TILE_COUNT] = {
    {128, 128, 256, 1.00f}, {64, 128, 256, 0.95f}, {32, 128, 256, 0.80f}, {64, 64, 256, 0.85f},
    {32, 64, 128, 0.65f},   {32, 32, 64, 0.45f},   {128, 32, 256, 0.80f}, {64, 32, 128, 0.65f},
};

// Picks the tile shape and the split-K factor.  Large tiles reuse operands best; the deep layers of the
// encoder (6x8 / 12x16 maps, K up to 4608) and the dense layers have too few output tiles to fill 256
// CUs, so their K loop is cut into slices until about two workgroups per CU exist.
ConvPlan choose_conv_plan(int Mpad, long pixels, int nclasses, int Kpad, long ws_floats)
{
    const int nsteps = Kpad / 16;
    ConvPlan best{TILE_32x32, 1};
    float best_score = -1.0f;
    for (int t = 0; t < TILE_COUNT; ++t) {
        const TileInfo &ti = kTiles[t];
        if (Mpad % ti.bm) continue;
        if (ti.bn > 32 && pixels * 2 <= ti.bn) continue;  // more than half of the pixel tile would be padding
        const long wgs = (long)(Mpad / ti.bm) * ((pixels + ti.bn - 1) / ti.bn) * nclasses;
        int split = 1;
        if (wgs < 384) {
            split = (int)((512 + wgs - 1) / wgs);
            const int smax = nsteps / 4 > 1 ? nsteps / 4 : 1;
            if (split > smax) split = smax;
            while (split > 1 && (long)nclasses * split * Mpad * pixels > ws_floats) --split;
        }
        float fill = (float)(wgs * split) / 512.0f;
        if (fill > 1.0f) fill = 1.0f;
        const float score = ti.eff * fill / (1.0f + 0.12f * (split - 1));
        if (score > best_score) { best_score = score; best = ConvPlan{t, split}; }
    }
    return best;
}

int conv_tile_bm(int tile) { return kTiles[tile].bm; }
int conv_tile_bn(int tile) { return kTiles[tile].bn; }

template <int BM, int BN, int WM, int WN>
static void launch_tile(const ConvArgs &a, dim3 grid, hipStream_t stream)
{
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN>), grid, dim3(64 * WM * WN), 0, stream, a);
}

void launch_conv_mfma(const ConvArgs &a_in, ConvPlan plan, int nclasses, hipStream_t stream)
{
    ConvArgs a = a_in;
    a.ksplit = plan.ksplit;
    const long P = (long)a.N * a.Hp * a.Wp;
    const TileInfo ti = kTiles[plan.tile];
    dim3 grid((unsigned)((P + ti.bn - 1) / ti.bn), (unsigned)(a.Mpad / ti.bm), (unsigned)(nclasses * plan.ksplit));
    switch (plan.tile) {
        case TILE_128x128: launch_tile<128, 128, 2, 2>(a, grid, stream); break;
        case TILE_64x128:  launch_tile<64, 128, 2, 2>(a, grid, stream); break;
        case TILE_32x128:  launch_tile<32, 128, 1, 4>(a, grid, stream); break;
        case TILE_64x64:   launch_tile<64, 64, 2, 2>(a, grid, stream); break;
        case TILE_32x64:   launch_tile<32, 64, 1, 2>(a, grid, stream); break;
        case TILE_128x32:  launch_tile<128, 32, 4, 1>(a, grid, stream); break;
        case TILE_64x32:   launch_tile<64, 32, 2, 1>(a, grid, stream); break;
        default:           launch_tile<32, 32, 1, 1>(a, grid, stream); break;
    }
    if (plan.ksplit > 1) launch_splitk_reduce(a, nclasses, stream);
}

thread_local int g_active_cus = 0;
thread_local hipEvent_t g_reduce_mark = nullptr;
thread_local bool g_reduce_marked = false;

void launch_splitk_reduce(const ConvArgs &a, int nclasses, hipStream_t stream)
{
    const long P = (long)a.N * a.Hp * a.Wp;
    if (g_reduce_mark && hipEventRecord(g_reduce_mark, stream) == hipSuccess) g_reduce_marked = true;
    // diagnostic hook (tools/ablate_lanes.py, row "split-K reduces"): DEMON_SKIP_REDUCE=1 leaves every split-K reduce launch out -- WRONG
    // results, used only to read off what the reduce launches cost with several passes in flight
    static const bool skip = [] {
        const char *e = getenv("DEMON_SKIP_REDUCE");
        const bool on = e && *e && atoi(e);
        if (on) fprintf(stderr, "libdemon_hip: DEMON_SKIP_REDUCE is set -- split-K reduce launches are LEFT OUT, all results are WRONG (diagnostic for tools/ablate_lanes.py only)\n");
        return on;
    }();
    if (skip) return;
    if (nclasses == 1 && a.osx == 1 && a.osy == 1 && (a.Wp & 3) == 0 && (a.Wo & 3) == 0 && (a.out_plane & 3) == 0 && (a.out_n_stride & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {
        dim3 rgrid((unsigned)((P / 4 + 255) / 256), (unsigned)a.Cout, 1);
        hipLaunchKernelGGL(conv_splitk_reduce4_kernel, rgrid, dim3(256), 0, stream, a);
        return;
    }
    dim3 rgrid((unsigned)((P + 255) / 256), (unsigned)a.Cout, (unsigned)nclasses);
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, rgrid, dim3(256), 0, stream, a);
}

}  // namespace demon
