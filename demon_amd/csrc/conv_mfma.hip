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

    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int cls = blockIdx.z;
    const int m0 = blockIdx.y * BM;
    const long p0 = (long)blockIdx.x * BN;
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

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < BPER; ++i) {
            const KEntry e = ktab[k0 + bg + i * BROWS];
            const int dy = e.dydx >> 16;
            const int dx = (int)(short)(e.dydx & 0xffff);
            const bool ok = (unsigned)(iy0 + dy) < (unsigned)a.H && (unsigned)(ix0 + dx) < (unsigned)a.W;
            breg[i] = ok ? inb[e.delta] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < APER; ++i) {
            const int row = arow + i * AROWS;
            if (AROWS * APER == BK || row < BK)
                areg[i] = *reinterpret_cast<const floatx4 *>(wp + (long)(k0 + row) * a.Mpad + m0 + ac4 * 4);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BPER; ++i) Bs[buf][bg + i * BROWS][bj] = breg[i];
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

    const int nsteps = a.Kpad / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_tiles((s + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + lhi;
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = As[buf][k][(wm * TM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = Bs[buf][k][(wn * TN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nsteps) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, leaky relu, optional per-sample scale of channel 0, coalesced NCHW store
    const int pyc = (gridDim.z > 1) ? (cls >> 1) : 0, pxc = (gridDim.z > 1) ? (cls & 1) : 0;
    const long plane = (long)a.Ho * a.Wo;
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

struct TileInfo { int bm, bn, threads; };
static const TileInfo kTiles[TILE_COUNT] = {
    {128, 128, 256}, {64, 128, 256}, {32, 128, 256}, {64, 64, 256}, {32, 64, 128}, {32, 32, 64},
};

int choose_conv_tile(int Mpad, long pixels, int nclasses)
{
    // largest tile (most operand reuse) that still yields >= 1.5 workgroups per CU; otherwise the
    // candidate with the most workgroups (small feature maps: 6x8 and 12x16 levels, dense layers)
    static const int order[TILE_COUNT] = {TILE_128x128, TILE_64x128, TILE_64x64, TILE_32x128, TILE_32x64, TILE_32x32};
    int best = -1;
    long best_wgs = -1;
    for (int oi = 0; oi < TILE_COUNT; ++oi) {
        const int t = order[oi];
        if (Mpad % kTiles[t].bm) continue;
        const long wgs = (long)(Mpad / kTiles[t].bm) * ((pixels + kTiles[t].bn - 1) / kTiles[t].bn) * nclasses;
        if (wgs >= 384) return t;
        if (wgs > best_wgs) { best_wgs = wgs; best = t; }
    }
    return best;
}

void launch_conv_mfma(const ConvArgs &a, int tile, int nclasses, hipStream_t stream)
{
    const long P = (long)a.N * a.Hp * a.Wp;
    const TileInfo ti = kTiles[tile];
    dim3 grid((unsigned)((P + ti.bn - 1) / ti.bn), (unsigned)(a.Mpad / ti.bm), (unsigned)nclasses);
    dim3 block(ti.threads);
    switch (tile) {
        case TILE_128x128: hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 2, 2>), grid, block, 0, stream, a); break;
        case TILE_64x128:  hipLaunchKernelGGL((conv_mfma_kernel<64, 128, 2, 2>), grid, block, 0, stream, a); break;
        case TILE_32x128:  hipLaunchKernelGGL((conv_mfma_kernel<32, 128, 1, 4>), grid, block, 0, stream, a); break;
        case TILE_64x64:   hipLaunchKernelGGL((conv_mfma_kernel<64, 64, 2, 2>), grid, block, 0, stream, a); break;
        case TILE_32x64:   hipLaunchKernelGGL((conv_mfma_kernel<32, 64, 1, 2>), grid, block, 0, stream, a); break;
        default:           hipLaunchKernelGGL((conv_mfma_kernel<32, 32, 1, 1>), grid, block, 0, stream, a); break;
    }
}

}  // namespace demon
