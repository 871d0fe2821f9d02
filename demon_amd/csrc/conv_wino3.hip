// conv_wino3.hip -- 3 x 3 stride-1 convolutions (the 24-channel heads, blocks_original.py:23-51, :238-294; the refinement net's conv1_1 /
// conv2_1 and the 64 -> 16 conv of its depth head, :484-511) as three 1 x 3 minimal-filtering row filters with the TRANSFORMED INPUT ROWS
// KEPT STATIONARY.
//
// conv_wino.hip's wino1d kernel runs such a layer as three passes over the channels, one per kernel row ky: every pass stages the
// input row y + ky - 1 of its tiles, transforms it (F(2,3) along x: 4 values from a window of 4 pixels) and multiplies it by U[ky].  An
// input row is therefore loaded, transformed and written to LDS three times -- once for each of the output rows y - 1, y, y + 1 it
// contributes to -- and every MFMA needs a B operand of its own from LDS.  Here a wave owns TN CONSECUTIVE output rows of 16 tiles
// (32 pixels) each: per K-step it reads the TN + 2 transformed input rows of its columns once and feeds each of them to up to three
// MFMAs (one per ky, into the accumulators of three different output rows):
//       acc[row r][e] += U[ky][e] (A operand) x T[input row r + ky - 1][e] (B operand),   ky = 0, 1, 2
//   staging (loads, transform VALU, LDS writes) per MFMA: (TN + 2) / (3 TN) of the three-pass form   (TN = 4: one half)
//   LDS operand reads per MFMA: (TN + 2 + 3) / (3 TN) instead of (TN + 1) / TN                       (TN = 4: 0.75 instead of 1.25)
// which is what a layer with few output channels needs (a transformed input has few consumers there: one 16-channel block for the
// depth head, two for the 24-channel heads) -- fp32 MFMAs share the SIMD's issue time with the vector ALU (conv_wino.hip).
// F(2,3) form: same transformed weights as the wino1d kernel (U[ky][e][ci][co], wino1d_repack_kernel with cross = 3), same arithmetic
// per output up to the order in which the three rows' products are added (ky is now the inner index of a K-step, not the outer one).
// F(4,3) form (template parameter F4): FOUR consecutive outputs of a row per tile from a window of six pixels with 6 products per
// kernel row instead of 12 (F(2,3): 8) -- 0.5 x the direct convolution's multiply-adds; interpolation points 0, +-1, +-2, infinity
// (wino1d_tables.h, generated and checked in exact rationals; fp32 error 2 x a direct fp32 sum at K = 128, tools/gen_wino1d.py).
// Its weights U43[ky][e][ci][co] (6 planes per kernel row) live in a buffer of their own (wino3_repack43_kernel).
// Stride-2 form (FORM 2; the refinement net's conv1 / conv2, blocks_original.py:484-511): output row r reads the input rows 2r - 1, 2r,
// 2r + 1, so a wave's TN output rows share 2 TN + 1 transformed input rows (the odd ones feed two output rows); along x the polyphase
// split gives the even / odd input samples the even / odd taps as stride-1 filters: four outputs from a window of nine pixels with
// F(4,2) + F(4,1) = 9 products per kernel row instead of 12 (0.75 x the direct convolution's multiply-adds, transforms of the even
// samples only).  Its weights U[ky][e][ci][co] (9 planes per kernel row) come from the same repack kernel.
#include <type_traits>

#include "internal.h"
#include "wino1d_tables.h"

namespace demon {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ int w3div(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }
}

struct Rows23 {   // F(2,3) along x: the 3-tap stride-1 kind of wino1d_tables.h with the interface of Wino43
    static constexpr int NUV = Wino1D<0>::NUV, WIN = Wino1D<0>::WIN, OUT = 2;
    static __device__ __forceinline__ void input(const float (&d)[WIN], float (&t)[NUV]) { Wino1D<0>::input(d, t); }
    template <class T> static __device__ __forceinline__ void output(const T (&m)[NUV], T (&o)[OUT]) { Wino1D<0>::output(m, o[0], o[1]); }
};

// WM x WN waves: WM 16-channel blocks x WN blocks of 16 tile columns; TN output rows per wave; KG K groups (of 4 channels) per barrier;
// FORM 0: stride 1, tiles of two pixels (F(2,3)); 1: stride 1, tiles of four pixels (F(4,3)); 2: stride 2, tiles of four pixels (F(4,2) + F(4,1))
template <int WM, int WN, int TN, int KG, bool MASK, int FORM>
__global__ __launch_bounds__(64 * WM * WN, 2) void wino3_rows_kernel(Wino3Args a)   // (2 waves per SIMD: 256 registers; 3 spills -- the F(4,3) shapes therefore hold 2 or 3 rows per wave: 24 accumulator registers per row)
{
    using K = typename std::conditional<FORM == 2, Wino4K3S2, typename std::conditional<FORM == 1, Wino43, Rows23>::type>::type;
    constexpr bool F4 = FORM == 1, S2 = FORM == 2;
    constexpr int SY = S2 ? 2 : 1;   // stride (rows and columns)
    constexpr int NUV = K::NUV, WIN = K::WIN, OUT = K::OUT, NT = 64 * WM * WN, CKS = 4 * KG;
    constexpr int PWD = S2 ? 9 : 6;   // floats a unit loads: F(2,3) three 8-byte vectors around its window, F(4,3) exactly its six pixels, stride 2 its nine
    constexpr int BM = 16 * WM, TCOLS = 16 * WN, RIN = SY * (TN - 1) + 3;
    constexpr int SLOTS = RIN * TCOLS;                         // (input row, tile column) slots per (e, channel)
    constexpr int TP = SLOTS + ((SLOTS & 31) ? 0 : 16);        // pitch: the k = 0 / 1 halves of a 32-lane LDS access on different banks
    constexpr int NUNIT = SLOTS * CKS;                         // staging units (channel, slot) per K-step
    constexpr int UNITS = (NUNIT + NT - 1) / NT;
    constexpr int NE = 3 * NUV;                                // (ky, e) weight planes
    constexpr int ASZ = NE * CKS * BM, TSZ = NUV * CKS * TP;
    constexpr int A4 = NE * BM * KG;                           // 16-byte chunks of the weight tile
    constexpr int APER = (A4 + NT - 1) / NT;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // As[2][ASZ], Ts[2][TSZ], one dummy 16-byte slot per thread

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lk = lane >> 4;
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
    const int m0 = by * BM;
    const int t1 = w3div((int)bx, a.m_colsx);
    const int cb = (int)bx - t1 * a.cols_x;           // column block
    const int n = w3div(t1, a.m_rowsy);
    const int rb = t1 - n * a.rows_y;                  // row block
    const int r0 = rb * TN, c0 = cb * TCOLS;           // first output row / first tile column
    const float *__restrict__ in0 = a.in + (long)n * a.in_n_stride;
    const int HW = a.H * a.W;

    // ---- staging units: unit i of this thread = (channel k of the K-step, input row j, tile column t).  F(2,3): the window of tile
    // column c = c0 + t (4 pixels from x = 2 c - 1) lies inside the three 8-byte vectors [2 (c - 1), 2 (c + 2)) of row r0 - 1 + j;
    // F(4,3): the window is the pixel 4 c - 1, the 16-byte vector [4 c, 4 c + 4) and the pixel 4 c + 4
    int goff[UNITS][3], tw[2][UNITS];
    unsigned lastmask = 0;
    const int last_c0 = (a.csteps - 1) * CKS;
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
        const int w = tid + i * NT;
        const bool uv = NUNIT % NT == 0 || w < NUNIT;
        const int k = uv ? w / SLOTS : 0, slot = uv ? w - k * SLOTS : 0;
        const int j = slot / TCOLS, t = slot - j * TCOLS;
        const int gy = SY * r0 - 1 + j, c = c0 + t;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            // stride 2: the window is the pixel 8 c - 1 and the 16-byte vectors [8 c, 8 c + 4), [8 c + 4, 8 c + 8)
            const int gx = S2 ? (e == 0 ? 8 * c - 1 : (e == 1 ? 8 * c : 8 * c + 4)) : (F4 ? (e == 0 ? 4 * c - 1 : (e == 1 ? 4 * c : 4 * c + 4)) : 2 * (c - 1 + e));
            const bool ok = uv & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
            goff[i][e] = ok ? 4 * (k * HW + gy * a.W + gx) : OOB;
        }
        tw[0][i] = uv ? 2 * ASZ + k * TP + slot : 2 * ASZ + 2 * TSZ + tid * 4;   // (units past the end write all their values to the first word of the thread's dummy slot)
        tw[1][i] = uv ? tw[0][i] + TSZ : tw[0][i];
        asm volatile("" : "+v"(tw[1][i]));
        lastmask |= ((last_c0 + k < a.Cin) ? 1u : 0u) << i;
    }
    // ---- weight loader: chunk f of the [ky * NUV + e][channel block][k][16] tile <-> U[ky][e][c0 + k][m0 + 16 blk + 4 c4 ..]
    int aoff[APER], aw[2][APER];
#pragma unroll
    for (int i = 0; i < APER; ++i) {
        const int f = tid + i * NT;
        const bool fv = A4 % NT == 0 || f < A4;
        const int c4 = f & 3, k = (f >> 2) % CKS, blk = (f / (4 * CKS)) % WM, e = f / (4 * CKS * WM);
        aoff[i] = fv ? 4 * (int)(((long)e * a.Cin4 + k) * a.Mpad + m0 + blk * 16 + c4 * 4) : OOB;
        aw[0][i] = fv ? f * 4 : 2 * ASZ + 2 * TSZ + tid * 4;
        aw[1][i] = fv ? f * 4 + ASZ : 2 * ASZ + 2 * TSZ + tid * 4;
    }
    int ra[2], rt[2];
    ra[0] = wm * (16 * CKS) + lane;
    ra[1] = ra[0] + ASZ;
    rt[0] = 2 * ASZ + lk * TP + wn * 16 + l15;
    rt[1] = rt[0] + TSZ;
    asm volatile("" : "+v"(ra[1]));
    asm volatile("" : "+v"(rt[1]));

    // extents of the two operand streams from THIS workgroup's base (clamped once, internal.h: rsrc_bytes); a K-step only subtracts its advance
    const int in_bytes0 = rsrc_bytes(view_floats_left(a.N, n, a.in_n_stride, a.Cin, 0, HW, HW));
    const int w_bytes0 = rsrc_bytes((NE * a.Cin4 + kWinoWeightSlackRows) * a.Mpad);

    floatx4 acc[TN][NUV];
#pragma unroll
    for (int tb = 0; tb < TN; ++tb)
#pragma unroll
        for (int e = 0; e < NUV; ++e) acc[tb][e] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float pregA[UNITS][PWD], pregB[UNITS][PWD];
    floatx4 aregA[APER], aregB[APER];
    auto load_tiles = [&](float (&preg)[UNITS][PWD], floatx4 (&areg)[APER], int cs) {
        const auto prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in0 + (long)cs * CKS * HW), 0, in_bytes0 - 4 * cs * CKS * HW, 0x00020000);
        const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wu + (long)cs * CKS * a.Mpad), 0, w_bytes0 - 4 * cs * CKS * a.Mpad, 0x00020000);
        // channels past Cin (last K-step, Cin not a multiple of 4 KG) are NOT read -- the planes behind the last channel of the last
        // image may lie behind the end of the allocation: their units load from the out-of-range offset, i.e. zeros
        auto units = [&](auto last_step) {   // (the masked form only in the one K-step that needs it: no extra VALU in the others)
            constexpr bool LAST = decltype(last_step)::value;
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                const int dead = (LAST && !((lastmask >> i) & 1u)) ? OOB : 0;   // (or-ed into the offsets: OOB covers every bit of a valid offset)
                if constexpr (S2) {
                    preg[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, goff[i][0] | dead, 0, 0));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, goff[i][1 + h] | dead, 0, 0));
#pragma unroll
                        for (int j = 0; j < 4; ++j) preg[i][1 + 4 * h + j] = v[j];
                    }
                } else if constexpr (F4) {
                    preg[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, goff[i][0] | dead, 0, 0));
                    const floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, goff[i][1] | dead, 0, 0));
#pragma unroll
                    for (int j = 0; j < 4; ++j) preg[i][1 + j] = v[j];
                    preg[i][5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, goff[i][2] | dead, 0, 0));
                } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(prsrc, goff[i][e] | dead, 0, 0));
                        preg[i][2 * e] = v[0];
                        preg[i][2 * e + 1] = v[1];
                    }
                }
            }
        };
        if (MASK && cs == a.csteps - 1) units(std::true_type{});
        else units(std::false_type{});
#pragma unroll
        for (int i = 0; i < APER; ++i) areg[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[i], 0, 0));
    };
    auto transform_store = [&](const float (&preg)[UNITS][PWD], const floatx4 (&areg)[APER], int buf, int cs) {
        (void)cs;   // (channels past Cin were loaded as zeros: their transformed values are zeros)
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            float d[WIN], t[NUV];
#pragma unroll
            for (int e = 0; e < WIN; ++e) d[e] = preg[i][(FORM == 0 ? 1 : 0) + e];   // the window starts one pixel left of the tile
            K::input(d, t);
            float *T = smem + tw[buf][i];
            if constexpr (NUNIT % NT == 0) {
#pragma unroll
                for (int e = 0; e < NUV; ++e) T[e * CKS * TP] = t[e];
            } else {
                const int st = (tid + i * NT < NUNIT) ? CKS * TP : 0;   // (units past the end: every value to the one dummy word)
#pragma unroll
                for (int e = 0; e < NUV; ++e) T[e * st] = t[e];
            }
        }
#pragma unroll
        for (int i = 0; i < APER; ++i) *reinterpret_cast<floatx4 *>(smem + aw[buf][i]) = areg[i];
    };
    // one item = one (K group, e): 3 weight fragments (ky) + RIN transformed input rows -> 3 TN MFMAs; the LDS reads of item g + 1 are
    // issued before the MFMAs of item g (see conv_wino.hip: with few waves per SIMD the matrix pipe otherwise idles for an LDS latency)
    constexpr int NI = KG * NUV;
    auto compute = [&](int buf) {
        const float *A = smem + ra[buf];
        const float *T = smem + rt[buf];
        float af[2][3], tf[2][RIN];
        auto fetch = [&](int it, int set) {
            const int kg = it / NUV, e = it - kg * NUV;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) af[set][ky] = A[(ky * NUV + e) * (WM * 16 * CKS) + kg * 64];
#pragma unroll
            for (int j = 0; j < RIN; ++j) tf[set][j] = T[(e * CKS + kg * 4) * TP + j * TCOLS];
        };
        fetch(0, 0);
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            if (it + 1 < NI) fetch(it + 1, (it + 1) & 1);
            __builtin_amdgcn_sched_barrier(0x16);
            const int e = it % NUV;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb)
                    acc[tb][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[it & 1][ky], tf[it & 1][SY * tb + ky], acc[tb][e], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x16);
        }
    };

    const int nsteps = a.csteps;
    auto phys = [&](int x) { return min(x, nsteps - 1); };   // a run-ahead past the end re-reads the last step
    load_tiles(pregA, aregA, phys(0));
    load_tiles(pregB, aregB, phys(1));
    transform_store(pregA, aregA, 0, phys(0));
    __syncthreads();
    {
        int s = 0;
        for (; s + 2 < nsteps; s += 2) {
            load_tiles(pregA, aregA, phys(s + 2));
            compute(0);
            transform_store(pregB, aregB, 1, phys(s + 1));
            __syncthreads();
            load_tiles(pregB, aregB, phys(s + 3));
            compute(1);
            transform_store(pregA, aregA, 0, phys(s + 2));
            __syncthreads();
        }
        if (s + 1 < nsteps) {
            compute(0);
            transform_store(pregB, aregB, 1, phys(s + 1));
            __syncthreads();
            compute(1);
        } else {
            compute(0);
        }
    }

    // ---- epilogue: the two outputs of a tile from its NUV accumulators; lane = tile column, registers = 4 consecutive channels
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (long)n * a.out_n_stride + (long)m0 * a.out_plane, 0, rsrc_bytes(view_floats_left(a.N, n, a.out_n_stride, a.Cout, m0, a.out_plane, (long)a.Ho * a.Wo)), 0x00020000);
    const int plane4 = 4 * (int)a.out_plane;
    const int x0 = OUT * (c0 + wn * 16 + l15);
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int y = r0 + tb;
        const bool tv = y < a.Ho && x0 < a.Wo;
        const int toff = tv ? 4 * (y * a.Wo + x0) + (wm * 16 + 4 * lk) * plane4 : OOB;
        // the output transform on the four channels of an accumulator register group at once (round 6): packed fp32 instructions, two channels each
        floatx4 m4[NUV], o4[OUT];
#pragma unroll
        for (int e = 0; e < NUV; ++e) m4[e] = acc[tb][e];
        K::output(m4, o4);
        const floatx4 b4 = *reinterpret_cast<const floatx4 *>(a.bias + m0 + wm * 16 + 4 * lk);   // (bias is padded to Mpad, a multiple of 16)
        const float slope = a.act ? 0.1f : 1.0f;   // leaky relu as max(v, slope v), branch-free (slope 1: the identity)
#pragma unroll
        for (int j = 0; j < OUT; ++j) {
            o4[j] += b4;
            o4[j] = __builtin_elementwise_max(o4[j], slope * o4[j]);
        }
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const int col = wm * 16 + 4 * lk + e4;
            const int off = (tv && m0 + col < a.Cout) ? toff + e4 * plane4 : OOB;
            if constexpr (OUT == 4) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, floatx4{o4[0][e4], o4[1][e4], o4[2][e4], o4[3][e4]}), orsrc, off, 0, 0);
            } else {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{o4[0][e4], o4[1][e4]}), orsrc, off, 0, 0);
            }
        }
    }
}

// U[ky][e][ci][co] = sum_t G[e][t] wp[(ky*3 + t)*Cin + ci][co] (K = Wino43: F(4,3); Wino4K3S2: stride 2); rows ci >= Cin stay zero (the buffer is zero-filled once)
template <class K>
__global__ __launch_bounds__(256) void wino3_repack43_kernel(float *__restrict__ wu, const float *__restrict__ wp, int Cin, int Cin4, int Mpad)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Cin * Mpad) return;
    const int ci = (int)(idx / Mpad), co = (int)(idx - (long)ci * Mpad);
    for (int ky = 0; ky < 3; ++ky) {
        float w[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) w[t] = wp[((long)(ky * 3 + t) * Cin + ci) * Mpad + co];
#pragma unroll
        for (int e = 0; e < K::NUV; ++e) {
            float u = 0.0f;
#pragma unroll
            for (int t = 0; t < 3; ++t) u += K::g(e, t) * w[t];
            wu[(((long)ky * K::NUV + e) * Cin4 + ci) * Mpad + co] = u;
        }
    }
}

void launch_wino3_repack43(float *wu, const float *wp, int Cin, int Cin4, int Mpad, int stride, hipStream_t s)
{
    const long total = (long)Cin * Mpad;
    if (stride == 2) hipLaunchKernelGGL(wino3_repack43_kernel<Wino4K3S2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wu, wp, Cin, Cin4, Mpad);
    else hipLaunchKernelGGL(wino3_repack43_kernel<Wino43>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wu, wp, Cin, Cin4, Mpad);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// variants 0 .. 7: F(2,3) tiles of two pixels; 8 .. 15: F(4,3) tiles of four pixels; 16 .. 19: the stride-2 form (tiles of four pixels)
struct W3Shape { int wm, wn, tn, kg; };
constexpr int W3_SHAPES = 8, W3_S2_SHAPES = 4;
static_assert(WINO3_VARIANTS == 2 * W3_SHAPES + W3_S2_SHAPES, "eight workgroup shapes per stride-1 form, four for stride 2");
static const W3Shape kW3Shapes[WINO3_VARIANTS] = {{2, 2, 4, 1}, {2, 4, 4, 1}, {4, 1, 4, 1}, {4, 2, 4, 1}, {1, 4, 4, 1}, {1, 4, 2, 2}, {2, 2, 2, 2}, {4, 1, 2, 2},
                                                  {2, 2, 2, 2}, {4, 1, 2, 2}, {2, 2, 3, 1}, {4, 1, 3, 1}, {1, 4, 2, 1}, {1, 4, 3, 1}, {2, 1, 2, 2}, {4, 2, 3, 1},
                                                  {4, 1, 2, 1}, {2, 2, 2, 1}, {4, 2, 2, 1}, {8, 1, 2, 1}};
static const W3Shape &w3shape(int v) { return kW3Shapes[v]; }
int wino3_variant_form(int v) { return v < W3_SHAPES ? 0 : (v < 2 * W3_SHAPES ? 1 : 2); }
bool wino3_variant_f4(int v) { return wino3_variant_form(v) == 1; }
int wino3_variant_bm(int v) { return 16 * w3shape(v).wm; }
int wino3_variant_kg(int v) { return w3shape(v).kg; }
int wino3_variant_rows(int v) { return w3shape(v).tn; }
int wino3_variant_cols(int v) { return 16 * w3shape(v).wn; }   // tile columns (2 or 4 pixels each) per workgroup

static size_t wino3_lds_bytes(int v)
{
    const W3Shape s = w3shape(v);
    const int form = wino3_variant_form(v);
    const int nuv = form == 2 ? 9 : (form == 1 ? 6 : 4), rin = (form == 2 ? 2 : 1) * (s.tn - 1) + 3;
    const int cks = 4 * s.kg, slots = rin * 16 * s.wn, tp = slots + ((slots & 31) ? 0 : 16);
    return sizeof(float) * (2ul * (3 * nuv * cks * 16 * s.wm + nuv * cks * tp) + 4ul * 64 * s.wm * s.wn);
}

bool wino3_plan_geometry(Wino3Args &a, int variant)
{
    if (variant < 0 || variant >= WINO3_VARIANTS) return false;
    const W3Shape s = w3shape(variant);
    const int form = wino3_variant_form(variant);
    const int out = form == 0 ? 2 : 4;                                 // pixels per tile
    if ((form == 2) != (a.stride == 2)) return false;
    if (form == 2 ? (a.W % 8 || a.Wo != a.W / 2 || a.Ho != (a.H + 1) / 2) : (a.Wo != a.W || a.Ho != a.H)) return false;
    if (a.Wo % out || a.Mpad % (16 * s.wm)) return false;              // windows are read as 8- / 16-byte vectors
    if (s.wm == 1 && a.Cout > 16) return false;                        // (one channel block per workgroup is for <= 16 channels)
    if (16 * s.wm > 16 && a.Cout <= 16) return false;
    if (a.Wo < 32) return false;                                       // narrower maps stay on the wino1d kernel (its tiles span images)
    a.rows_y = (a.Ho + s.tn - 1) / s.tn;
    a.cols_x = (a.Wo / out + 16 * s.wn - 1) / (16 * s.wn);
    if ((double)a.rows_y * s.tn * a.cols_x * 16 * out * s.wn > 2.5 * a.Ho * a.Wo) return false;   // mostly empty tile slots: not worth measuring
    a.csteps = (a.Cin + 4 * s.kg - 1) / (4 * s.kg);
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    a.m_colsx = magic(a.cols_x);
    a.m_rowsy = magic(a.rows_y);
    return wino3_lds_bytes(variant) <= 160 * 1024;
}

long wino3_workgroups(const Wino3Args &a, int variant)
{
    return (long)a.N * a.rows_y * a.cols_x * ((a.Cout + wino3_variant_bm(variant) - 1) / wino3_variant_bm(variant));
}

template <int WM, int WN, int TN, int KG, bool MASK, int FORM>
static bool launch_w3m(const Wino3Args &a, dim3 grid, size_t lds, hipStream_t s)
{
    static PerDeviceOnce once;
    if (lds > 48 * 1024 &&
        !once.ensure([] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&wino3_rows_kernel<WM, WN, TN, KG, MASK, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }))
        return false;
    hipLaunchKernelGGL((wino3_rows_kernel<WM, WN, TN, KG, MASK, FORM>), grid, dim3(64 * WM * WN), lds, s, a);
    return true;
}

template <int WM, int WN, int TN, int KG, int FORM>
static bool launch_w3(const Wino3Args &a, dim3 grid, size_t lds, hipStream_t s)
{
    if (a.Cin % (4 * KG)) return launch_w3m<WM, WN, TN, KG, true, FORM>(a, grid, lds, s);
    return launch_w3m<WM, WN, TN, KG, false, FORM>(a, grid, lds, s);
}

// a.wu: U[ky][e][Cin4][Mpad] with 4 planes per kernel row for variants 0 .. 7, U43 with 6 planes for variants 8 .. 15, 9 planes for 16 .. 19
bool launch_wino3(const Wino3Args &a, int variant, hipStream_t stream)
{
    const int bm = wino3_variant_bm(variant);
    dim3 grid((unsigned)(a.N * a.rows_y * a.cols_x), (unsigned)((a.Cout + bm - 1) / bm), 1);
    const size_t lds = wino3_lds_bytes(variant);
    switch (variant) {
        case 0: return launch_w3<2, 2, 4, 1, 0>(a, grid, lds, stream);
        case 1: return launch_w3<2, 4, 4, 1, 0>(a, grid, lds, stream);
        case 2: return launch_w3<4, 1, 4, 1, 0>(a, grid, lds, stream);
        case 3: return launch_w3<4, 2, 4, 1, 0>(a, grid, lds, stream);
        case 4: return launch_w3<1, 4, 4, 1, 0>(a, grid, lds, stream);
        case 5: return launch_w3<1, 4, 2, 2, 0>(a, grid, lds, stream);
        case 6: return launch_w3<2, 2, 2, 2, 0>(a, grid, lds, stream);
        case 7: return launch_w3<4, 1, 2, 2, 0>(a, grid, lds, stream);
        case 8: return launch_w3<2, 2, 2, 2, 1>(a, grid, lds, stream);
        case 9: return launch_w3<4, 1, 2, 2, 1>(a, grid, lds, stream);
        case 10: return launch_w3<2, 2, 3, 1, 1>(a, grid, lds, stream);
        case 11: return launch_w3<4, 1, 3, 1, 1>(a, grid, lds, stream);
        case 12: return launch_w3<1, 4, 2, 1, 1>(a, grid, lds, stream);
        case 13: return launch_w3<1, 4, 3, 1, 1>(a, grid, lds, stream);
        case 14: return launch_w3<2, 1, 2, 2, 1>(a, grid, lds, stream);
        case 15: return launch_w3<4, 2, 3, 1, 1>(a, grid, lds, stream);
        case 16: return launch_w3<4, 1, 2, 1, 2>(a, grid, lds, stream);
        case 17: return launch_w3<2, 2, 2, 1, 2>(a, grid, lds, stream);
        case 18: return launch_w3<4, 2, 2, 1, 2>(a, grid, lds, stream);
        default: return launch_w3<8, 1, 2, 1, 2>(a, grid, lds, stream);
    }
}

}  // namespace demon
