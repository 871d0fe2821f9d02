// conv_frag.hip -- LDS-tiled fp32-MFMA contraction on fragment-ordered weights, for the deep small-map layers.
//
// conv_stream.hip removed LDS and barriers from these layers and found the next limit: operand bytes through the vector
// memory path (an ablation and a prefetch-depth sweep, DESIGN.md section 3.1: neither deeper prefetch nor removing the waits
// helps; removing the loads does).  With every wave streaming its own A and B, a 64 x 32 wave tile moves 384 operand bytes per
// MFMA.  Here the WM x WN waves of a workgroup share both operands through LDS, so a 128 x 64 tile moves 192 bytes per MFMA,
// and the K loop keeps the pipeline of conv_patch.hip: one barrier per K-step, placed BETWEEN the two halves of the step's
// MFMAs, fragments of the next half read behind the current one, global loads two steps ahead.
//   A: the fragment-order copy of the weights (conv_stream.hip: Wf[cls][step][32-row block][lane][8]).  The workgroup's A tile of
//      a K-step is ONE contiguous chunk of BM * 16 floats: copied with 16-byte loads, stored to LDS as [block][half][lane][4] so that
//      a lane fetches the 4 + 4 values of its 8 MFMA groups with two conflict-free ds_read_b128;
//   B: a thread owns one pixel of the tile and 16 * BN / NT reduction rows of the step: 4-byte loads from the NCHW activations (taps
//      outside the image read the page of zeros), LDS [16][BN], fragments by ds_read_b32 (lane = pixel, half-wave = k parity).
// Requirements as conv_stream.hip (Cin % 16 == 0).  Reduction order per output: tap major, channel minor == conv_mfma's order.
#include <type_traits>

#include "internal.h"

namespace demon {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// WM x WN waves, a wave owns 32*TM channels x 32*TN pixels: BM = 32*TM*WM, BN = 32*TN*WN.  KS = K-steps (of 16 reduction
// indices) per pipeline stage, i.e. per barrier: 2 for the small wave tiles, whose 8 MFMAs per step are too few between barriers.
// KW > 1: KW groups of WM x WN waves share the output tile and split its K range (each with its own LDS buffers, all meeting at the
// same barriers); their accumulators are summed through LDS in group order at the end -- split-K without partial sums in HBM and
// without a reduce launch, for the layers whose tiles are too few to fill the chip.
template <int WM, int WN, int TM, int TN, int KS, int KW>
__device__ __forceinline__ void frag_tile(const StreamArgs &s)
{
    const ConvArgs &a = s.c;
    TlScope tl(a.tl);
    constexpr int NT = 64 * WM * WN;   // threads of one wave group
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int A4 = BM * 4;                  // float4 chunks of the A tile of one K-step
    constexpr int APER = (A4 + NT - 1) / NT;
    constexpr int BPER = 16 * BN / NT;          // reduction rows per thread
    static_assert((A4 % NT == 0 || NT % A4 == 0) && (16 * BN) % NT == 0 && NT % BN == 0, "bad tile");

    constexpr int NG = 8 * KS, G1 = NG / 2;     // MFMA groups per stage / per half
    constexpr int ASZ = KS * BM * 16, BSZ = KS * 16 * BN;   // floats of one A / B buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int gk = threadIdx.x / NT;                        // K-splitting group of this thread
    float *const Asg = smem + gk * (2 * (ASZ + BSZ));       // [buf][sub-step][block][half][lane][4]
    float *const Bsg = Asg + 2 * ASZ;                       // [buf][k][pixel]

    const int tid = threadIdx.x - gk * NT, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int cls = blockIdx.z / a.ksplit;
    const int zs = blockIdx.z - cls * a.ksplit;
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
    const int m0 = by * BM;
    const long p0 = (long)bx * BN;
    const long P = (long)a.N * a.Hp * a.Wp;
    const int HW = a.H * a.W;

    // ---- B staging: this thread's pixel and its BPER rows of every K-step
    const int bpx = tid % BN, brow0 = (tid / BN) * BPER;
    const float *__restrict__ pb;
    unsigned okbits = 0;
    {
        const long p = p0 + bpx;
        const bool pv = p < P;
        const long pc = pv ? p : 0;
        const int x = (int)(pc % a.Wp);
        const long t = pc / a.Wp;
        const int y = (int)(t % a.Hp);
        const int n = (int)(t / a.Hp);
        const int iy0 = y * a.sy, ix0 = x * a.sx;
        pb = a.in + (long)n * a.in_n_stride + (long)iy0 * a.W + ix0 + (long)brow0 * HW;
        for (int t2 = 0; t2 < s.ntaps; ++t2) {
            const bool ok = pv & ((unsigned)(iy0 + s.tapdy[cls][t2]) < (unsigned)a.H) & ((unsigned)(ix0 + s.tapdx[cls][t2]) < (unsigned)a.W);
            okbits |= (ok ? 1u : 0u) << t2;
        }
    }
    // ---- A staging: float4 chunk q = tid + i*NT of the contiguous tile; global chunk (block, lane, half) -> LDS (block, half, lane)
    const bool a_thread = A4 >= NT || tid < A4;   // BM = 32: the A tile has fewer 16-byte chunks than the workgroup has threads
    int alds[APER];
#pragma unroll
    for (int i = 0; i < APER; ++i) {
        const int q = (tid + i * NT) % A4;
        const int blk = q >> 7, r = q & 127, ln = r >> 1, half = r & 1;
        alds[i] = (blk * 128 + half * 64 + ln) * 4;
    }

    const int per_slice = (s.nsteps + a.ksplit - 1) / a.ksplit;
    int s_begin = zs * per_slice;
    int s_end = min(s.nsteps, s_begin + per_slice);
    int nsteps = max(s_end - s_begin, 0);
    if (KW > 1) {  // this group's share; every group runs the same number of stages (missing steps load zeros) to meet the same barriers
        const int per_group = (nsteps + KW - 1) / KW;
        s_begin += gk * per_group;
        s_end = min(s_end, s_begin + per_group);
        nsteps = per_group;
    }
    const long a_step = (long)(a.Mpad >> 5) * 512;
    const float *__restrict__ wf = s.wf + cls * s.cls_wf_stride + (long)s_begin * a_step + (long)(m0 >> 5) * 512 + (tid % A4) * 4;

    // loader position: step ls = (tap lt, channel block lc); cur = this thread's B address of channel row brow0 of that step
    int ls = s_begin;
    int lt = s_begin / s.csteps, lc = s_begin - lt * s.csteps;
    const long cstride = (long)16 * HW;
    const float *__restrict__ cur;
    auto enter_tap = [&]() {
        const int t2 = min(lt, s.ntaps - 1);
        const int tdelta = s.tapdy[cls][t2] * a.W + s.tapdx[cls][t2];
        cur = (((okbits >> t2) & 1u) ? pb + tdelta : s.zero + (long)brow0 * HW) + (long)lc * cstride;
    };
    enter_tap();

    floatx4 areg[2][KS][APER];
    float breg[2][KS][BPER];
    auto load = [&](int set) {   // one stage = KS K-steps; steps past the end of the slice become zeros (they multiply to nothing)
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            if (ls < s_end) {  // (wave-uniform)
#pragma unroll
                for (int i = 0; i < APER; ++i) areg[set][u][i] = *reinterpret_cast<const floatx4 *>(wf + (long)i * NT * 4);
#pragma unroll
                for (int i = 0; i < BPER; ++i) breg[set][u][i] = cur[(long)i * HW];
                wf += a_step;
                cur += cstride;
                ++ls;
                if (++lc == s.csteps) { lc = 0; ++lt; enter_tap(); }
            } else {
#pragma unroll
                for (int i = 0; i < APER; ++i) areg[set][u][i] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < BPER; ++i) breg[set][u][i] = 0.0f;
            }
        }
    };
    // piece q of a stage's LDS stores: q < KS*APER -> an A chunk, else a B element
    auto store_piece = [&](int set, int q, int buf) {
        if (q < KS * APER) {
            const int u = q / APER, i = q - u * APER;
            if (a_thread) *reinterpret_cast<floatx4 *>(&Asg[buf * ASZ + u * (BM * 16) + alds[i]]) = areg[set][u][i];
        } else {
            const int r = q - KS * APER, u = r / BPER, i = r - u * BPER;
            Bsg[buf * BSZ + (u * 16 + brow0 + i) * BN + bpx] = breg[set][u][i];
        }
    };
    constexpr int NPIECE = KS * (APER + BPER);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragments of one stage: group g = sub-step g / 8, k pair g % 8.  ah[q][i] = the 4 A values of groups 4q .. 4q+3 of row block i
    floatx4 ah[2 * KS][TM];
    float bv[NG][TN];
    auto read_group = [&](int buf, int g) {   // the B values of group g, and with the first group of every four also their A values
        if ((g & 3) == 0) {
            const int u = g >> 3, half = (g >> 2) & 1;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                ah[g >> 2][i] = *reinterpret_cast<const floatx4 *>(&Asg[buf * ASZ + u * (BM * 16) + (((wm * TM + i) * 2 + half) * 64 + lane) * 4]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[g][j] = Bsg[buf * BSZ + (2 * g + lhi) * BN + (wn * TN + j) * 32 + l31];
    };
    auto mfma_group = [&](int g) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[g >> 2][i][g & 3], bv[g][j], acc[i][j], 0, 0, 0);
    };
    // one stage on buffer `buf` (see conv_patch.hip for the buffer-safety argument of the mid-stage barrier):
    //   groups [0,G1)  | reads of groups [G1,NG) of this stage | global loads of stage t+2 into set `lset` | LDS writes of stage t+1 from `sset`
    //   barrier
    //   groups [G1,NG) | reads of groups [0,G1) of stage t+1
    auto kstage = [&](int buf, int lset, int sset, auto loads, auto store) {
        constexpr bool LOADS = decltype(loads)::value, STORE = decltype(store)::value;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            mfma_group(g);
            read_group(buf, G1 + g);
            if (LOADS && g == 1) load(lset);
            if (STORE) {
#pragma unroll
                for (int q = g * NPIECE / G1; q < (g + 1) * NPIECE / G1; ++q) store_piece(sset, q, buf ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STORE) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = G1; g < NG; ++g) {
            mfma_group(g);
            if (STORE) read_group(buf ^ 1, g - G1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int nstages = (nsteps + KS - 1) / KS;
    if (nstages > 0) {
        load(0);                 // stage 0
        load(1);                 // stage 1 (zeros when the slice has one stage)
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) store_piece(0, q, 0);
    }
    __syncthreads();
    tl.mark(1);
    if (nstages > 0) {
#pragma unroll
        for (int g = 0; g < G1; ++g) read_group(0, g);
    }
    // stage t on buffer t&1: loads of stage t+2 into the set stage t came from; the other set (stage t+1) goes to the other buffer
    int st = 0;
    for (; st + 2 < nstages; st += 2) {
        kstage(0, 0, 1, std::true_type{}, std::true_type{});
        kstage(1, 1, 0, std::true_type{}, std::true_type{});
    }
    if (st + 1 < nstages) {
        kstage(0, 0, 1, std::false_type{}, std::true_type{});
        kstage(1, 0, 0, std::false_type{}, std::false_type{});
    } else if (nstages > 0) {
        kstage(0, 0, 0, std::false_type{}, std::false_type{});
    }
    tl.mark(2);

    if constexpr (KW > 1) {
        // sum the KW partial tiles through LDS (the tile buffers are free now): every group parks its accumulators, group g adds
        // up the registers r = g, g + KW, ... of all groups in group order, group 0 collects the sums and runs the epilogue alone
        constexpr int R = TM * TN * 16;
        __syncthreads();
        float *base = smem + (long)wave * (R * 64) + lane;
        float *mine = base + (long)gk * (WM * WN * R * 64);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
        __syncthreads();
        for (int r = gk; r < R; r += KW) {
            float sum = base[r * 64];
            for (int g = 1; g < KW; ++g) sum += base[(long)g * (WM * WN * R * 64) + r * 64];
            base[r * 64] = sum;
        }
        __syncthreads();
        if (gk != 0) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = base[((i * TN + j) * 16 + r) * 64];
    }

    // ---- epilogue (as conv_mfma.hip)
    const int mw = m0 + wm * TM * 32;
    const long pw = p0 + (long)wn * TN * 32;
    if (a.ksplit > 1) {  // raw partial sums to the workspace [cls][slice][Mpad][P]; conv_splitk_reduce finishes
        float *__restrict__ ws = a.ws + ((long)blockIdx.z * a.Mpad) * P;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long p = pw + j * 32 + l31;
            if (p >= P) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = mw + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    ws[(long)co * P + p] = acc[i][j][r];
                }
        }
        return;
    }
    const int pyc = cls >> 1, pxc = cls & 1;
    const long plane = a.out_plane;
    if (a.osx == 1 && a.osy == 1 && (a.Wp & 3) == 0 && (a.Cout & 3) == 0 && a.scale == nullptr) {
        const int q = l31 >> 2, li = lane & 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long p = pw + j * 32 + 4 * q;
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
                const int co = mw + 32 * i + li + 8 * rb + 4 * lhi;
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
        const long p = pw + j * 32 + l31;
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
                const int co = mw + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (co < a.Cout) {
                    float v = acc[i][j][r] + a.bias[co];
                    if (a.act) v = v >= 0.0f ? v : 0.1f * v;
                    if (co == 0) v *= sc;
                    ob[(long)co * plane] = v;
                }
            }
    }
}

template <int WM, int WN, int TM, int TN, int KS, int KW>
__global__ __launch_bounds__(64 * WM * WN * KW) void conv_frag_kernel(StreamArgs s)
{
    frag_tile<WM, WN, TM, TN, KS, KW>(s);
}

// Two dependent layers in ONE launch: the k x 1 and the 1 x k conv of a stride-1 separable pair (helpers.py:105-153), when a
// workgroup's tile holds ALL output channels (BM == Mpad) of WHOLE image rows (BN a multiple of the row length).  The 1 x k conv
// of those rows then reads nothing but what this workgroup's k x 1 conv just wrote (same rows, columns +-k/2, every channel), so the
// pair needs no grid-wide dependency: first layer, its stores drained and visible to the workgroup (they went through this CU's L1),
// barrier, second layer on the same tile.  Same kernels, same order of operations as the two launches -> the same bits; what
// disappears is a launch boundary with its drain / refill of the whole chip, and the workgroups stop marching in lock-step.
template <int WM, int WN, int TM, int TN, int KS>
__global__ __launch_bounds__(64 * WM * WN) void conv_frag_chain_kernel(StreamArgs s1, StreamArgs s2)
{
    frag_tile<WM, WN, TM, TN, KS, 1>(s1);
    __syncthreads();   // workgroup-scope release / acquire around the barrier: the first layer's stores are complete and visible
    frag_tile<WM, WN, TM, TN, KS, 1>(s2);
}


struct FragVariant { int wm, wn, tm, tn, ks, kw; };
static const FragVariant kFragVariants[FRAG_VARIANTS] = {
    {2, 2, 2, 1, 1, 1}, {2, 2, 1, 1, 1, 1}, {2, 2, 2, 2, 1, 1}, {2, 2, 1, 2, 1, 1}, {4, 1, 2, 1, 1, 1}, {1, 4, 2, 1, 1, 1}, {4, 1, 1, 1, 1, 1},
    {1, 4, 2, 2, 1, 1}, {2, 2, 1, 1, 2, 1}, {4, 1, 1, 1, 2, 1}, {2, 2, 2, 1, 2, 1}, {1, 4, 2, 1, 2, 1}, {2, 2, 1, 2, 2, 1}, {1, 4, 1, 1, 2, 1},
    {4, 1, 1, 1, 1, 2}, {4, 1, 1, 1, 1, 4}, {2, 2, 1, 1, 1, 2}, {2, 2, 1, 1, 1, 4}, {2, 2, 2, 1, 1, 2}, {2, 2, 2, 1, 1, 4}, {1, 4, 2, 1, 1, 2}, {4, 1, 2, 1, 1, 2}};

int frag_variant_bm(int v) { return 32 * kFragVariants[v].tm * kFragVariants[v].wm; }
int frag_variant_bn(int v) { return 32 * kFragVariants[v].tn * kFragVariants[v].wn; }
int frag_variant_kw(int v) { return kFragVariants[v].kw; }

template <int WM, int WN, int TM, int TN, int KS, int KW>
static void launch_frag_instance(const StreamArgs &s, dim3 grid, hipStream_t stream)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr size_t tiles = sizeof(float) * KW * 2 * (KS * BM * 16 + KS * 16 * BN);
    constexpr size_t red = KW > 1 ? sizeof(float) * KW * WM * WN * TM * TN * 16 * 64 : 0;
    constexpr size_t lds = tiles > red ? tiles : red;
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KB of LDS");
    if (lds > 48 * 1024) {  // more dynamic LDS than the default limit: opt in once per instantiation and device
        static PerDeviceOnce once;
        once.ensure([] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_frag_kernel<WM, WN, TM, TN, KS, KW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess; });
    }
    hipLaunchKernelGGL((conv_frag_kernel<WM, WN, TM, TN, KS, KW>), grid, dim3(64 * WM * WN * KW), lds, stream, s);
}

template <int WM, int WN, int TM, int TN, int KS>
static void launch_frag_chain_instance(const StreamArgs &s1, const StreamArgs &s2, dim3 grid, hipStream_t stream)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr size_t lds = sizeof(float) * 2 * (KS * BM * 16 + KS * 16 * BN);
    if (lds > 48 * 1024) {
        static PerDeviceOnce once;
        once.ensure([] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_frag_chain_kernel<WM, WN, TM, TN, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess; });
    }
    hipLaunchKernelGGL((conv_frag_chain_kernel<WM, WN, TM, TN, KS>), grid, dim3(64 * WM * WN), lds, stream, s1, s2);
}

// true when the pair ran as one chained launch; false (nothing launched) when the variant / shapes do not allow it
bool launch_conv_frag_chain(const StreamArgs &s1_in, const StreamArgs &s2_in, int variant, hipStream_t stream)
{
    if (variant < 0 || variant >= FRAG_VARIANTS || kFragVariants[variant].kw != 1) return false;
    StreamArgs s1 = s1_in, s2 = s2_in;
    s1.c.ksplit = s2.c.ksplit = 1;
    const ConvArgs &a = s1.c, &b = s2.c;
    const int bm = frag_variant_bm(variant), bn = frag_variant_bn(variant);
    const bool same_grid = a.N == b.N && a.Hp == b.Hp && a.Wp == b.Wp && a.Mpad == b.Mpad;
    if (!same_grid || a.Mpad != bm || bn % a.Wp != 0 || b.in != a.out || b.sy != 1 || b.sx != 1 || a.osx != 1 || a.osy != 1 || b.osx != 1 || b.osy != 1) return false;
    // the second conv may only look sideways (1 x k): rows other than the tile's own belong to other workgroups
    for (int t = 0; t < s2.ntaps; ++t)
        if (s2.tapdy[0][t] != 0) return false;
    const long P = (long)a.N * a.Hp * a.Wp;
    dim3 grid((unsigned)((P + bn - 1) / bn), 1, 1);
    switch (variant) {
        case 0: launch_frag_chain_instance<2, 2, 2, 1, 1>(s1, s2, grid, stream); break;
        case 1: launch_frag_chain_instance<2, 2, 1, 1, 1>(s1, s2, grid, stream); break;
        case 2: launch_frag_chain_instance<2, 2, 2, 2, 1>(s1, s2, grid, stream); break;
        case 3: launch_frag_chain_instance<2, 2, 1, 2, 1>(s1, s2, grid, stream); break;
        case 4: launch_frag_chain_instance<4, 1, 2, 1, 1>(s1, s2, grid, stream); break;
        case 5: launch_frag_chain_instance<1, 4, 2, 1, 1>(s1, s2, grid, stream); break;
        case 6: launch_frag_chain_instance<4, 1, 1, 1, 1>(s1, s2, grid, stream); break;
        case 7: launch_frag_chain_instance<1, 4, 2, 2, 1>(s1, s2, grid, stream); break;
        case 8: launch_frag_chain_instance<2, 2, 1, 1, 2>(s1, s2, grid, stream); break;
        case 9: launch_frag_chain_instance<4, 1, 1, 1, 2>(s1, s2, grid, stream); break;
        case 10: launch_frag_chain_instance<2, 2, 2, 1, 2>(s1, s2, grid, stream); break;
        case 11: launch_frag_chain_instance<1, 4, 2, 1, 2>(s1, s2, grid, stream); break;
        case 12: launch_frag_chain_instance<2, 2, 1, 2, 2>(s1, s2, grid, stream); break;
        default: launch_frag_chain_instance<1, 4, 1, 1, 2>(s1, s2, grid, stream); break;
    }
    return true;
}

void launch_conv_frag(const StreamArgs &s_in, int variant, int ksplit, int nclasses, hipStream_t stream)
{
    StreamArgs s = s_in;
    s.c.ksplit = ksplit;
    const ConvArgs &a = s.c;
    const long P = (long)a.N * a.Hp * a.Wp;
    const int bm = frag_variant_bm(variant), bn = frag_variant_bn(variant);
    dim3 grid((unsigned)((P + bn - 1) / bn), (unsigned)(a.Mpad / bm), (unsigned)(nclasses * ksplit));
    switch (variant) {
        case 0: launch_frag_instance<2, 2, 2, 1, 1, 1>(s, grid, stream); break;
        case 1: launch_frag_instance<2, 2, 1, 1, 1, 1>(s, grid, stream); break;
        case 2: launch_frag_instance<2, 2, 2, 2, 1, 1>(s, grid, stream); break;
        case 3: launch_frag_instance<2, 2, 1, 2, 1, 1>(s, grid, stream); break;
        case 4: launch_frag_instance<4, 1, 2, 1, 1, 1>(s, grid, stream); break;
        case 5: launch_frag_instance<1, 4, 2, 1, 1, 1>(s, grid, stream); break;
        case 6: launch_frag_instance<4, 1, 1, 1, 1, 1>(s, grid, stream); break;
        case 7: launch_frag_instance<1, 4, 2, 2, 1, 1>(s, grid, stream); break;
        case 8: launch_frag_instance<2, 2, 1, 1, 2, 1>(s, grid, stream); break;
        case 9: launch_frag_instance<4, 1, 1, 1, 2, 1>(s, grid, stream); break;
        case 10: launch_frag_instance<2, 2, 2, 1, 2, 1>(s, grid, stream); break;
        case 11: launch_frag_instance<1, 4, 2, 1, 2, 1>(s, grid, stream); break;
        case 12: launch_frag_instance<2, 2, 1, 2, 2, 1>(s, grid, stream); break;
        case 13: launch_frag_instance<1, 4, 1, 1, 2, 1>(s, grid, stream); break;
        case 14: launch_frag_instance<4, 1, 1, 1, 1, 2>(s, grid, stream); break;
        case 15: launch_frag_instance<4, 1, 1, 1, 1, 4>(s, grid, stream); break;
        case 16: launch_frag_instance<2, 2, 1, 1, 1, 2>(s, grid, stream); break;
        case 17: launch_frag_instance<2, 2, 1, 1, 1, 4>(s, grid, stream); break;
        case 18: launch_frag_instance<2, 2, 2, 1, 1, 2>(s, grid, stream); break;
        case 19: launch_frag_instance<2, 2, 2, 1, 1, 4>(s, grid, stream); break;
        case 20: launch_frag_instance<1, 4, 2, 1, 1, 2>(s, grid, stream); break;
        default: launch_frag_instance<4, 1, 2, 1, 1, 2>(s, grid, stream); break;
    }
    if (ksplit > 1) launch_splitk_reduce(s.c, nclasses, stream);
}

}  // namespace demon
