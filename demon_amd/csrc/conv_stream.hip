// conv_stream.hip -- register-streaming fp32-MFMA contraction for the deep, small-map layers (and the dense layers).
//
// The 6x8 / 12x16 / 24x32 maps of the encoders (conv3_1 ... conv5_1, refine4/upconv; helpers.py:105-153,
// blocks_original.py:97-110) are GEMMs with few pixels (N = 1.5 k ... 25 k at batch 32), many channels and K = 384 ... 2560:
// M x N / (32 x 32) output blocks barely outnumber the 1024 SIMDs, so the LDS-tiled kernels run them as 128 x 32 tiles with
// ONE accumulator block per wave and a barrier every 8 MFMAs -- per K-step a wave sits out the LDS read latency, the wait for
// its own gathers and the barrier, and the matrix pipe idles half of the K loop (tools/timeline.py: 30 us of K loop for
// 15 us of MFMA work; SQ_VALU_MFMA_BUSY 40 %).
//
// Here a wave never touches LDS and never meets a barrier:
//   * A (weights) is re-packed once per weight upload into MFMA FRAGMENT ORDER, Wf[cls][step][m-tile][lane][8]: the 8 A values a
//     lane feeds into the 8 MFMAs of a K-step (16 reduction indices = 16 channels of one tap) are 32 contiguous bytes, the
//     wave's fragment 2 KB contiguous: two 16-byte loads per lane per step, no address arithmetic, nothing shared between waves;
//   * B (activations, NCHW) goes straight from global memory into the MFMA operand register: lane (pixel, k parity) loads
//     in[n][c0 + 2kk + parity][y*sy + dy][x*sx + dx]; the 4 waves of a workgroup (stacked along Cout) hit the same lines (L1);
//     zero padding = a per-lane, per-tap validity bit decided once per kernel: taps outside the image read a page of zeros;
//   * loads run two K-steps ahead of the MFMAs in three rotating register sets, so a wave always has 16 MFMAs (~1000 cycles)
//     of its own work between issuing a load and using it, and waves drift apart instead of marching in lock-step.
// Requirements: Cin % 16 == 0 (a K-step never straddles a tap); same packed-weight source, epilogue, split-K workspace and
// reduce kernel as conv_mfma.hip.  Reduction order per output: tap major, channel minor == conv_mfma's order, so results
// are bit-identical to the im2col kernel for equal split-K.
#include <type_traits>

#include "internal.h"

namespace demon {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// d_wp [cls][Krows][Mpad] (row k = tap*Cin + ci) -> Wf [cls][step][Mpad/32][64][8]:
//   Wf[...][lane = 32*hi + r][kk] = W[k = 16*step + 2*kk + hi][m = 32*mtile + r]
// grid: (K/16 * Mpad/32, ncls), 64 threads
__global__ __launch_bounds__(64) void stream_repack_kernel(float *__restrict__ wf, const float *__restrict__ wp, int nsteps, int mtiles,
                                                           int Mpad, long cls_w_stride, long cls_wf_stride)
{
    const int cls = blockIdx.y;
    const int step = blockIdx.x / mtiles, mt = blockIdx.x - step * mtiles;
    const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
    const float *__restrict__ src = wp + cls * cls_w_stride + (long)(16 * step + hi) * Mpad + 32 * mt + r;
    floatx4 v0, v1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        v0[kk] = src[(long)(2 * kk) * Mpad];
        v1[kk] = src[(long)(2 * kk + 8) * Mpad];
    }
    float *__restrict__ dst = wf + cls * cls_wf_stride + (((long)step * mtiles + mt) * 64 + lane) * 8;
    *reinterpret_cast<floatx4 *>(dst) = v0;
    *reinterpret_cast<floatx4 *>(dst + 4) = v1;
}

void launch_stream_repack(float *wf, const float *wp, int ncls, int K, int Mpad, long cls_w_stride, hipStream_t s)
{
    const int nsteps = K / 16, mtiles = Mpad / 32;
    hipLaunchKernelGGL(stream_repack_kernel, dim3((unsigned)(nsteps * mtiles), (unsigned)ncls), dim3(64), 0, s, wf, wp, nsteps, mtiles, Mpad,
                       cls_w_stride, (long)K * Mpad);
}

template <int TM, int TN>
struct StreamSet {
    floatx4 a0[TM], a1[TM];  // A fragments of the 8 MFMA groups of one K-step, per 32-channel row block
    float b[TN][8];          // B fragments, per 32-pixel column block
};

// NW waves stacked along Cout (BM = 32 * TM * NW), a wave owns 32*TM channels x 32*TN pixels: TM*TN accumulator blocks, and per
// K-step 8*TM*TN MFMAs on 2*TM 16-byte + 8*TN 4-byte loads (512 / 384 / 256 operand bytes per MFMA for 1x1 / 1x2 / 2x2).
// KW > 1 (small batches: few output tiles, long K): KW groups of NW waves share the tile and split its K range between them;
// their partial accumulators are summed through LDS at the end, in wave order -- split-K without partial sums in HBM and
// without a reduce launch.
template <int NW, int TM, int TN, int KW>
__device__ __forceinline__ void stream_tile(const StreamArgs &s)
{
    const ConvArgs &a = s.c;
    TlScope tl(a.tl);
    constexpr int BM = 32 * TM * NW, BN = 32 * TN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) % NW, wk = (tid >> 6) / NW;  // position along Cout / K slice inside the workgroup
    const int l31 = lane & 31, lhi = lane >> 5;
    const int cls = blockIdx.z / a.ksplit;
    const int zs = blockIdx.z - cls * a.ksplit;
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
    const int m0 = by * BM + wave * 32 * TM;  // this wave's first output channel
    const long p0 = (long)bx * BN;
    const long P = (long)a.N * a.Hp * a.Wp;
    const int HW = a.H * a.W;

    // ---- B addressing: lane = (pixel l31 of column block j, k parity lhi)
    const float *__restrict__ pb[TN];
    unsigned okbits[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long p = p0 + j * 32 + l31;
        const bool pv = p < P;
        const long pc = pv ? p : 0;
        const int x = (int)(pc % a.Wp);
        const long t = pc / a.Wp;
        const int y = (int)(t % a.Hp);
        const int n = (int)(t / a.Hp);
        const int iy0 = y * a.sy, ix0 = x * a.sx;
        pb[j] = a.in + (long)n * a.in_n_stride + (long)iy0 * a.W + ix0 + (long)lhi * HW;
        unsigned bits = 0;
        for (int t2 = 0; t2 < s.ntaps; ++t2) {
            const bool ok = pv & ((unsigned)(iy0 + s.tapdy[cls][t2]) < (unsigned)a.H) & ((unsigned)(ix0 + s.tapdx[cls][t2]) < (unsigned)a.W);
            bits |= (ok ? 1u : 0u) << t2;
        }
        okbits[j] = bits;
    }

    // ---- K range of this slice, in steps of 16 reduction indices
    const int per_slice = (s.nsteps + a.ksplit - 1) / a.ksplit;
    int s_begin = zs * per_slice;
    int s_end = min(s.nsteps, s_begin + per_slice);
    if (KW > 1) {  // this wave group's share of the workgroup's K range
        const int per_wave = (max(s_end - s_begin, 0) + KW - 1) / KW;
        s_begin += wk * per_wave;
        s_end = min(s_end, s_begin + per_wave);
    }
    const long a_step = (long)(a.Mpad >> 5) * 512;  // floats between consecutive steps of Wf
    const float *__restrict__ wf = s.wf + cls * s.cls_wf_stride + (long)s_begin * a_step + ((long)(m0 >> 5) * 64 + lane) * 8;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // running position of the loader: step ls = (tap lt, channel block lc); cur[j] = this lane's B address for channel lc*16
    // of tap lt -- inside the image: anchor + tap offset; outside: the zero page (zero padding without a select in the MFMA chain)
    int ls = s_begin;
    int lt = s_begin / s.csteps, lc = s_begin - lt * s.csteps;
    const long cstride = (long)16 * HW;
    const float *__restrict__ cur[TN];
    auto enter_tap = [&]() {
        const int t2 = min(lt, s.ntaps - 1);
        const int tdelta = s.tapdy[cls][t2] * a.W + s.tapdx[cls][t2];
#pragma unroll
        for (int j = 0; j < TN; ++j) cur[j] = (((okbits[j] >> t2) & 1u) ? pb[j] + tdelta : s.zero + (long)lhi * HW) + (long)lc * cstride;
    };
    enter_tap();

    auto load = [&](StreamSet<TM, TN> &f, auto guarded) {
        if (!decltype(guarded)::value || ls < s_end) {
#ifdef DEMON_STREAM_DBG
            if (!(s.dbg & 1)) {
#endif
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // plain loads: the fragment stream of a 32-row block is re-read by every pixel tile of the XCD out of its L2
                // (non-temporal loads were measured 6 % slower end to end)
                f.a0[i] = *reinterpret_cast<const floatx4 *>(wf + i * 512);
                f.a1[i] = *reinterpret_cast<const floatx4 *>(wf + i * 512 + 4);
            }
#ifdef DEMON_STREAM_DBG
            }
            if (!(s.dbg & 2))
#endif
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) f.b[j][kk] = cur[j][(long)(2 * kk) * HW];
                cur[j] += cstride;
            }
            wf += a_step;
            ++ls;
            if (++lc == s.csteps) {  // next tap (wave-uniform, once per Cin / 16 steps)
                lc = 0;
                ++lt;
                enter_tap();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](const StreamSet<TM, TN> &f) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float av = kk < 4 ? f.a0[i][kk & 3] : f.a1[i][kk & 3];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f.b[j][kk], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    };

#ifndef DEMON_STREAM_SETS
#define DEMON_STREAM_SETS 3
#endif
    constexpr int NS = DEMON_STREAM_SETS;  // register sets: loads run NS - 1 K-steps ahead of the MFMAs
    StreamSet<TM, TN> f[NS];
#ifdef DEMON_STREAM_DBG
    for (int q = 0; q < NS; ++q) {
        for (int i = 0; i < TM; ++i) f[q].a0[i] = f[q].a1[i] = floatx4{1.0f, 0.5f, 0.25f, 2.0f};
        for (int j = 0; j < TN; ++j)
            for (int kk = 0; kk < 8; ++kk) f[q].b[j][kk] = 0.001f * (lane + kk);
    }
#endif
#pragma unroll
    for (int q = 0; q < NS - 1; ++q) load(f[q], std::true_type{});
    tl.mark(1);
    int cs = s_begin;  // step being computed; f[q] holds step cs + q
    // steady state: every load of the iteration exists (steps cs+NS-1 .. cs+2NS-2), so the loop body has no branches
    while (cs + 2 * NS - 1 <= s_end) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            load(f[(q + NS - 1) % NS], std::false_type{});
            compute(f[q]);
        }
        cs += NS;
    }
    while (cs < s_end) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (cs + q < s_end) {
                load(f[(q + NS - 1) % NS], std::true_type{});
                compute(f[q]);
            }
        }
        cs += NS;
    }
    tl.mark(2);

    if constexpr (KW > 1) {
        // sum the KW partial tiles through LDS: every group parks its accumulators, then group g adds up the registers
        // r = g, g + KW, ... of all groups IN GROUP ORDER (results do not depend on timing) and hands the sums to group 0,
        // which runs the epilogue alone.  Two barriers, R + R / KW LDS reads per lane.
        constexpr int R = TM * TN * 16;
        extern __shared__ __attribute__((aligned(16))) float red[];   // [KW][NW][R][64]
        float *base = red + (long)wave * (R * 64) + lane;
        float *mine = base + (long)wk * (NW * R * 64);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
        __syncthreads();
        for (int r = wk; r < R; r += KW) {
            float sum = base[r * 64];
            for (int g = 1; g < KW; ++g) sum += base[(long)g * (NW * R * 64) + r * 64];
            base[r * 64] = sum;   // group 0's copy of register r (only this group touches register r now)
        }
        __syncthreads();
        if (wk != 0) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = base[((i * TN + j) * 16 + r) * 64];
    }

    // ---- epilogue (as conv_mfma.hip)
    if (a.ksplit > 1) {  // raw partial sums to the workspace [cls][slice][Mpad][P]; conv_splitk_reduce finishes
        float *__restrict__ ws = a.ws + ((long)blockIdx.z * a.Mpad) * P;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long p = p0 + j * 32 + l31;
            if (p >= P) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    ws[(long)co * P + p] = acc[i][j][r];
                }
        }
        return;
    }
    const int pyc = cls >> 1, pxc = cls & 1;
    const long plane = a.out_plane;
    if (a.osx == 1 && a.osy == 1 && (a.Wp & 3) == 0 && (a.Cout & 3) == 0 && a.scale == nullptr) {
        // 4x4 transpose inside every lane quad (two DPP stages): each lane stores 4 consecutive pixels of one channel (16 bytes)
        const int q = l31 >> 2, li = lane & 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long p = p0 + j * 32 + 4 * q;
            const bool ok = p < P;
            const long pc = ok ? p : 0;
            const int x = (int)(pc % a.Wp);
            const long t = pc / a.Wp;
            const int y = (int)(t % a.Hp);
            const int n = (int)(t / a.Hp);
            float *__restrict__ ob = a.out + (long)n * a.out_n_stride + (long)y * a.Wo + x;
#pragma unroll
            for (int irb = 0; irb < 4 * TM; ++irb) {
                const int i = irb >> 2, rb = irb & 3;
                float v0 = acc[i][j][4 * rb + 0], v1 = acc[i][j][4 * rb + 1], v2 = acc[i][j][4 * rb + 2], v3 = acc[i][j][4 * rb + 3];
                {
                    const bool odd = li & 1;
                    float s0 = odd ? v0 : v1, s1 = odd ? v2 : v3;
                    s0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
                    s1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
                    if (odd) { v0 = s0; v2 = s1; } else { v1 = s0; v3 = s1; }
                }
                {
                    const bool hi = li & 2;
                    float s0 = hi ? v0 : v2, s1 = hi ? v1 : v3;
                    s0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, true));
                    s1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, true));
                    if (hi) { v0 = s0; v1 = s1; } else { v2 = s0; v3 = s1; }
                }
                const int co = m0 + 32 * i + li + 8 * rb + 4 * lhi;
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
        const long p = p0 + j * 32 + l31;
        if (p >= P) continue;
        const int x = (int)(p % a.Wp);
        const long t = p / a.Wp;
        const int y = (int)(t % a.Hp);
        const int n = (int)(t / a.Hp);
        float *__restrict__ ob = a.out + (long)n * a.out_n_stride + (long)(y * a.osy + pyc) * a.Wo + (x * a.osx + pxc);
        const float sc = a.scale ? a.scale[n] : 1.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (co < a.Cout) {
                    float v = acc[i][j][r] + a.bias[co];
                    if (a.act) v = v >= 0.0f ? v : 0.1f * v;
                    if (co == 0) v *= sc;
                    ob[(long)co * plane] = v;
                }
            }
    }
}

template <int NW, int TM, int TN, int KW>
__global__ __launch_bounds__(64 * NW * KW) void conv_stream_kernel(StreamArgs s)
{
    stream_tile<NW, TM, TN, KW>(s);
}

// the k x 1 and the 1 x k conv of a stride-1 separable pair in one launch (see conv_frag_chain_kernel, conv_frag.hip): the NW waves
// of a workgroup hold all output channels of whole image rows; the barrier between the layers is the only one in this kernel
template <int NW, int TM, int TN>
__global__ __launch_bounds__(64 * NW) void conv_stream_chain_kernel(StreamArgs s1, StreamArgs s2)
{
    stream_tile<NW, TM, TN, 1>(s1);
    __syncthreads();
    stream_tile<NW, TM, TN, 1>(s2);
}


struct StreamVariant { int nw, tm, tn, kw; };
// (waves along Cout, 32-channel row blocks per wave, 32-pixel column blocks per wave, K-splitting wave groups per workgroup)
static const StreamVariant kStreamVariants[STREAM_VARIANTS] = {
    {4, 1, 1, 1}, {4, 1, 2, 1}, {2, 1, 1, 1}, {2, 1, 2, 1}, {1, 1, 1, 1}, {1, 1, 2, 1}, {2, 2, 2, 1}, {1, 2, 2, 1}, {2, 2, 1, 1}, {4, 2, 1, 1},
    {1, 1, 1, 4}, {1, 1, 1, 8}, {1, 1, 1, 16}, {1, 1, 2, 8}, {1, 2, 1, 8}, {1, 2, 2, 4}, {1, 2, 2, 8}, {2, 1, 1, 4}};

int stream_variant_bm(int v) { return 32 * kStreamVariants[v].nw * kStreamVariants[v].tm; }
int stream_variant_bn(int v) { return 32 * kStreamVariants[v].tn; }
int stream_variant_waves(int v) { return kStreamVariants[v].nw * kStreamVariants[v].kw; }
int stream_variant_kw(int v) { return kStreamVariants[v].kw; }

template <int NW, int TM, int TN, int KW>
static void launch_stream_variant(const StreamArgs &s, dim3 grid, hipStream_t stream)
{
    const size_t lds = KW > 1 ? sizeof(float) * KW * NW * TM * TN * 16 * 64 : 0;
    hipLaunchKernelGGL((conv_stream_kernel<NW, TM, TN, KW>), grid, dim3(64 * NW * KW), lds, stream, s);
}

bool launch_conv_stream_chain(const StreamArgs &s1_in, const StreamArgs &s2_in, int variant, hipStream_t stream)
{
    if (variant < 0 || variant >= STREAM_VARIANTS || kStreamVariants[variant].kw != 1) return false;
    StreamArgs s1 = s1_in, s2 = s2_in;
    s1.c.ksplit = s2.c.ksplit = 1;
    const ConvArgs &a = s1.c, &b = s2.c;
    const int bm = stream_variant_bm(variant), bn = stream_variant_bn(variant);
    const bool same_grid = a.N == b.N && a.Hp == b.Hp && a.Wp == b.Wp && a.Mpad == b.Mpad;
    if (!same_grid || a.Mpad != bm || bn % a.Wp != 0 || b.in != a.out || b.sy != 1 || b.sx != 1 || a.osx != 1 || a.osy != 1 || b.osx != 1 || b.osy != 1) return false;
    for (int t = 0; t < s2.ntaps; ++t)
        if (s2.tapdy[0][t] != 0) return false;
    const long P = (long)a.N * a.Hp * a.Wp;
    dim3 grid((unsigned)((P + bn - 1) / bn), 1, 1);
    switch (variant) {
        case 0: hipLaunchKernelGGL((conv_stream_chain_kernel<4, 1, 1>), grid, dim3(256), 0, stream, s1, s2); break;
        case 1: hipLaunchKernelGGL((conv_stream_chain_kernel<4, 1, 2>), grid, dim3(256), 0, stream, s1, s2); break;
        case 2: hipLaunchKernelGGL((conv_stream_chain_kernel<2, 1, 1>), grid, dim3(128), 0, stream, s1, s2); break;
        case 3: hipLaunchKernelGGL((conv_stream_chain_kernel<2, 1, 2>), grid, dim3(128), 0, stream, s1, s2); break;
        case 4: hipLaunchKernelGGL((conv_stream_chain_kernel<1, 1, 1>), grid, dim3(64), 0, stream, s1, s2); break;
        case 5: hipLaunchKernelGGL((conv_stream_chain_kernel<1, 1, 2>), grid, dim3(64), 0, stream, s1, s2); break;
        case 6: hipLaunchKernelGGL((conv_stream_chain_kernel<2, 2, 2>), grid, dim3(128), 0, stream, s1, s2); break;
        case 7: hipLaunchKernelGGL((conv_stream_chain_kernel<1, 2, 2>), grid, dim3(64), 0, stream, s1, s2); break;
        case 8: hipLaunchKernelGGL((conv_stream_chain_kernel<2, 2, 1>), grid, dim3(128), 0, stream, s1, s2); break;
        default: hipLaunchKernelGGL((conv_stream_chain_kernel<4, 2, 1>), grid, dim3(256), 0, stream, s1, s2); break;
    }
    return true;
}

void launch_conv_stream(const StreamArgs &s_in, int variant, int ksplit, int nclasses, hipStream_t stream)
{
    StreamArgs s = s_in;
    s.c.ksplit = ksplit;
    const ConvArgs &a = s.c;
    const long P = (long)a.N * a.Hp * a.Wp;
    const int bm = stream_variant_bm(variant), bn = stream_variant_bn(variant);
    dim3 grid((unsigned)((P + bn - 1) / bn), (unsigned)(a.Mpad / bm), (unsigned)(nclasses * ksplit));
    switch (variant) {
        case 0: launch_stream_variant<4, 1, 1, 1>(s, grid, stream); break;
        case 1: launch_stream_variant<4, 1, 2, 1>(s, grid, stream); break;
        case 2: launch_stream_variant<2, 1, 1, 1>(s, grid, stream); break;
        case 3: launch_stream_variant<2, 1, 2, 1>(s, grid, stream); break;
        case 4: launch_stream_variant<1, 1, 1, 1>(s, grid, stream); break;
        case 5: launch_stream_variant<1, 1, 2, 1>(s, grid, stream); break;
        case 6: launch_stream_variant<2, 2, 2, 1>(s, grid, stream); break;
        case 7: launch_stream_variant<1, 2, 2, 1>(s, grid, stream); break;
        case 8: launch_stream_variant<2, 2, 1, 1>(s, grid, stream); break;
        case 9: launch_stream_variant<4, 2, 1, 1>(s, grid, stream); break;
        case 10: launch_stream_variant<1, 1, 1, 4>(s, grid, stream); break;
        case 11: launch_stream_variant<1, 1, 1, 8>(s, grid, stream); break;
        case 12: launch_stream_variant<1, 1, 1, 16>(s, grid, stream); break;
        case 13: launch_stream_variant<1, 1, 2, 8>(s, grid, stream); break;
        case 14: launch_stream_variant<1, 2, 1, 8>(s, grid, stream); break;
        case 15: launch_stream_variant<1, 2, 2, 4>(s, grid, stream); break;
        case 16: launch_stream_variant<1, 2, 2, 8>(s, grid, stream); break;
        default: launch_stream_variant<2, 1, 1, 4>(s, grid, stream); break;
    }
    if (ksplit > 1) launch_splitk_reduce(s.c, nclasses, stream);
}

}  // namespace demon
