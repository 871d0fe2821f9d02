// conv_row.hip -- the 1 x k stride-(1,2) convs with at most 32 channels on both sides: `conv1x` (9 taps, the second half of the first
// layer of every block: blocks_original.py:141, :331 through helpers.py:105-153) and the iterative nets' `conv2x` (7 taps).  At batch 32
// `conv1x` moves 151 MB (33 us at 4.5 TB/s) for 28 us of MFMA time in the minimal-filtering form -- and has only 8 K-steps, so on the
// general 1-D kernel (conv_wino.hip, 74 us) prologue, LDS round trips of the transformed operands and epilogue outweigh the K loop.
//
// Same arithmetic as wino1d_kernel (wino1d_tables.h: two outputs per window with k + 2 instead of 2k products; the transformed weights
// U[e][ci][co] of wino1d_repack_kernel), other data flow:
//   * the whole reduction runs out of LDS: a workgroup stages U (NUV x Cin x 32 floats, 45 KB for 9 taps) and the raw input patch of
//     a 2-row x 64-column output tile (Cin x 2 x 136 floats, 35 KB) ONCE -- no K loop over global memory, one barrier per workgroup;
//   * a wave owns 16 tiles (32 outputs) x all 32 output channels, so the transformed input has exactly one consumer: a lane reads
//     the 12-pixel window of its (tile, channel) as three ds_read_b128, transforms it in registers and feeds the NUV values straight
//     into 2 x NUV MFMAs (16x16x4) as the B operand -- the T[e][k][tile] round trip through LDS of the general kernel is gone;
//   * A operands: U in LDS as [K-step][e][channel block][lane]: one register base per K-step + immediates, read one group of 4 e
//     ahead of their MFMAs (scheduling barriers pin the LDS / MFMA order, as in wino1d_kernel).
//   * workgroups are persistent (two per CU, 80 KB of LDS each): U is staged once per workgroup, and the global loads of the next
//     tile's patch are in flight (36 registers per lane) while the current one is computed; the second workgroup of the CU fills
//     the two barriers per tile.  (Measured on the way: one tile per workgroup, U reloaded by each -- 75 us, the same as the general
//     kernel; one persistent workgroup per CU with a double-buffered patch -- 81 us: with one wave per SIMD every LDS / barrier wait
//     is exposed, PMC: matrix pipe 39 %, vector ALU 24 %, waiting 34 % of the wave cycles.)
#include <type_traits>

#include "internal.h"
#include "wino1d_tables.h"

namespace demon {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int ROW_NT = 256, ROW_TX = 64, ROW_R = 2, ROW_PW = 136, ROW_MAXC = 32;

// CAFFE: the layer pads TAPS / 2 on the left (helpers.py's caffe padding); else (TAPS - 2) / 2, the 'same' padding of a stride-2 conv (v2)
template <int KIND, bool CAFFE>
__global__ __launch_bounds__(ROW_NT, 2) void conv_row_kernel(RowArgs a)
{
    using K = Wino1D<KIND>;
    constexpr int NUV = K::NUV, WIN = K::WIN, TAPS = K::TAPS;
    constexpr int NG = (NUV + 3) / 4;   // groups of 4 e
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int csteps = a.Cin4 >> 2;
    float *Us = smem;                               // [csteps][NUV][2][64]
    float *patch = smem + csteps * NUV * 128;       // [Cin4][2][136]
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int OOB = 0x7ffffff0;
    const int per_img = a.tiles_y * a.tiles_x, ntiles = a.N * per_img;
    const int patch_floats = a.Cin4 * ROW_R * ROW_PW;

    // ---- stage U once per workgroup (16-byte pieces; Mpad == 32: U[e][ci][co] is dense)
    constexpr int UPER = (NUV * ROW_MAXC * 8 + ROW_NT - 1) / ROW_NT, PPER = (ROW_MAXC * ROW_R * 34 + ROW_NT - 1) / ROW_NT;
    const int nu4 = NUV * a.Cin4 * 8, np4 = a.Cin4 * ROW_R * 34;
    {
        const auto ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wu), 0, rsrc_bytes((NUV * a.Cin4 + kWinoWeightSlackRows) * 32), 0x00020000);   // U[e][Cin4][Mpad = 32] + the slack rows
        floatx4 uv[UPER];
#pragma unroll
        for (int i = 0; i < UPER; ++i) {
            const int f = tid + i * ROW_NT;   // (e, ci, q): 4 output channels 4 q .. of U[e][ci]
            uv[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, f < nu4 ? 16 * f : OOB, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < UPER; ++i) {
            const int f = tid + i * ROW_NT;
            if (f < nu4) {
                // (Cin4 == 32: 256 pieces per e, i.e. e = i, ci = tid / 8 -- no division)
                const int q = f & 7, ci = a.Cin4 == ROW_MAXC ? ((tid >> 3) & 31) : (f >> 3) % a.Cin4, e = a.Cin4 == ROW_MAXC ? i : (f >> 3) / a.Cin4;
                *reinterpret_cast<floatx4 *>(Us + (((ci >> 2) * NUV + e) * 2 + (q >> 2)) * 64 + (ci & 3) * 16 + 4 * (q & 3)) = uv[i];
            }
        }
    }
    // ---- patch loader: piece f = tid + i * 256 of [Cin4 * 2 rows][34 x 4 columns], decoded once; per tile only the base moves
    int prow_off[PPER], pxq[PPER];   // byte offset of (ci, row) inside the image, first column of the piece
    bool prow_ok[PPER];
#pragma unroll
    for (int i = 0; i < PPER; ++i) {
        const int f = tid + i * ROW_NT;
        const int row = f / 34, xq = f - row * 34;   // row = ci * 2 + r
        const int ci = row >> 1;
        prow_ok[i] = f < np4 && ci < a.Cin;
        prow_off[i] = 4 * ((ci * a.H + (row & 1)) * a.W);
        pxq[i] = 4 * xq;
    }
    floatx4 pv[PPER];
    auto load_patch = [&](int t) {   // t = tile index (uniform); tiles past the end load nothing
        const int tc = min(t, ntiles - 1);
        const int n = tc / per_img, trem = tc - n * per_img;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int y0 = ty * ROW_R, xin0 = 2 * tx * ROW_TX - 4;   // input column of patch column 0 (a multiple of 4)
        const auto irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in + (long)n * a.in_n_stride + (long)y0 * a.W), 0, rsrc_bytes(view_floats_left(a.N, n, a.in_n_stride, a.Cin, 0, (long)a.H * a.W, (long)a.H * a.W) - y0 * a.W), 0x00020000);
#pragma unroll
        for (int i = 0; i < PPER; ++i) {
            const int gx = xin0 + pxq[i];
            const int row = (tid + i * ROW_NT) / 34;
            const bool ok = t < ntiles && prow_ok[i] && y0 + (row & 1) < a.H && gx >= 0 && gx < a.W;
            pv[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, ok ? prow_off[i] + 4 * gx : OOB, 0, 0));
        }
    };
    auto store_patch = [&](int buf) {
        float *p = patch + buf * patch_floats;   // (one buffer: buf == 0)
#pragma unroll
        for (int i = 0; i < PPER; ++i) {
            const int f = tid + i * ROW_NT;
            if (f < np4) *reinterpret_cast<floatx4 *>(p + 4 * f) = pv[i];   // [row][34 x 4]
        }
    };

    // ---- wave = (tile row r, 16-tile block tb); lane = (tile, channel of the K-step)
    const int r = wave >> 1, tile = (wave & 1) * 16 + l15;
    const int w_off = (lk * ROW_R + r) * ROW_PW + 4 * tile;   // + s * 4 * ROW_R * ROW_PW per K-step
    const float *A0 = Us + lane;                              // + s * NUV * 128 per K-step; + (e * 2 + cb) * 64

    load_patch((int)blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        store_patch(0);
        __syncthreads();
        load_patch(t + (int)gridDim.x);   // the next tile of this workgroup: in flight during the MFMAs below
        const float *W0 = patch + w_off;
        floatx4 acc[NUV][2];
        floatx4 wv[3];
        float af[2][4][2], afn[4][2];
        auto read_window = [&](const float *p) {
#pragma unroll
            for (int j = 0; j < 3; ++j) wv[j] = *reinterpret_cast<const floatx4 *>(p + 4 * j);
        };
        auto fetch = [&](const float *A, int g, float (&dst)[4][2]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * g + j;
                if (e >= NUV) continue;
                dst[j][0] = A[(e * 2 + 0) * 64];
                dst[j][1] = A[(e * 2 + 1) * 64];
            }
        };
        // one K-step = 4 input channels; FIRST: the accumulators start from the constant 0 operand instead of 88 zeroed registers
        auto kstep = [&](auto first, int s) {
            constexpr bool FIRST = decltype(first)::value;
            const float *A = A0 + s * (NUV * 128);
            const float *An = A0 + min(s + 1, csteps - 1) * (NUV * 128);
            float d[WIN], tt[NUV];
            if constexpr (CAFFE) {
#pragma unroll
                for (int e = 0; e < WIN; ++e) d[e] = wv[(4 - TAPS / 2 + e) >> 2][(4 - TAPS / 2 + e) & 3];
            } else {
#pragma unroll
                for (int e = 0; e < WIN; ++e) d[e] = wv[(4 - (TAPS - 2) / 2 + e) >> 2][(4 - (TAPS - 2) / 2 + e) & 3];
            }
            K::input(d, tt);
            read_window(W0 + min(s + 1, csteps - 1) * (4 * ROW_R * ROW_PW));   // consumed in the next K-step, behind 2 NUV MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j) { af[0][j][0] = afn[j][0]; af[0][j][1] = afn[j][1]; }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) fetch(A, g + 1, af[(g + 1) & 1]);
                else fetch(An, 0, afn);   // group 0 of the next K-step
                __builtin_amdgcn_sched_barrier(0x16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * g + j;
                    if (e >= NUV) continue;
                    const floatx4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                    acc[e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g & 1][j][0], tt[e], FIRST ? zero : acc[e][0], 0, 0, 0);
                    acc[e][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g & 1][j][1], tt[e], FIRST ? zero : acc[e][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0x16);
            }
        };
        read_window(W0);
        fetch(A0, 0, afn);
        kstep(std::true_type{}, 0);
        for (int s = 1; s < csteps; ++s) kstep(std::false_type{}, s);

        // ---- the two outputs of a tile from its NUV accumulators; lane = tile, registers = 4 consecutive channels; 8-byte stores
        const int n = t / per_img, trem = t - n * per_img;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (long)n * a.out_n_stride, 0, rsrc_bytes(view_floats_left(a.N, n, a.out_n_stride, a.Cout, 0, a.out_plane, (long)a.Ho * a.Wo)), 0x00020000);
        const int y = ty * ROW_R + r, x = tx * ROW_TX + 2 * tile;
        const bool tv = y < a.Ho && x < a.Wo;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            // the output transform on the four channels of an accumulator register group at once (round 6): packed fp32 instructions, two channels each
            floatx4 m4[NUV], p0, p1;
#pragma unroll
            for (int e = 0; e < NUV; ++e) m4[e] = acc[e][cb];
            K::output(m4, p0, p1);
            const floatx4 b4 = *reinterpret_cast<const floatx4 *>(a.bias + cb * 16 + 4 * lk);   // (padded to Mpad = 32)
            p0 += b4; p1 += b4;
            {
                const float slope = a.act ? 0.1f : 1.0f;   // leaky relu as max(v, slope v), branch-free (slope 1: the identity)
                p0 = __builtin_elementwise_max(p0, slope * p0);
                p1 = __builtin_elementwise_max(p1, slope * p1);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = cb * 16 + 4 * lk + q;
                const int off = (tv && co < a.Cout) ? 4 * (co * (int)a.out_plane + y * a.Wo + x) : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, floatx2{p0[q], p1[q]}), orsrc, off, 0, 0);
            }
        }
        __syncthreads();   // every wave is done reading the patch
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
bool conv_row_shape_ok(int kh, int kw, int sh, int sw, int ph, int pw, int Cin, int Mpad, int W, int Wo)
{
    if (kh != 1 || (kw != 7 && kw != 9) || sh != 1 || sw != 2 || ph != 0) return false;
    if (pw != kw / 2 && pw != (kw - 2) / 2) return false;
    return Cin >= 16 && (Cin + 3) / 4 * 4 <= ROW_MAXC && Mpad == 32 && (W & 3) == 0 && (Wo & 1) == 0 && Wo == (W + 1) / 2;
}

template <int KIND, bool CAFFE>
static bool launch_row_t(const RowArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    static PerDeviceOnce once;
    if (!once.ensure([] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_row_kernel<KIND, CAFFE>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess; })) return false;
    hipLaunchKernelGGL((conv_row_kernel<KIND, CAFFE>), grid, dim3(ROW_NT), lds, s, a);
    return true;
}

bool launch_conv_row(RowArgs a, int taps, hipStream_t stream)
{
    a.tiles_y = (a.Ho + ROW_R - 1) / ROW_R;
    a.tiles_x = (a.Wo + ROW_TX - 1) / ROW_TX;
    const int nuv = taps + 2;
    const size_t lds = sizeof(float) * ((size_t)(a.Cin4 / 4) * nuv * 128 + (size_t)a.Cin4 * ROW_R * ROW_PW);
    const int ntiles = a.N * a.tiles_y * a.tiles_x;
    // a workgroup walks `tpw` tiles (stride = grid size) and retires: U is staged once per tpw tiles, and other streams' workgroups
    // get a CU's LDS every few microseconds (fully persistent workgroups -- two per CU for the whole launch -- were 4 us faster alone
    // and gained nothing in the pipeline: the side stream's kernels could not start beside them)
    static const int tpw_env = getenv("DEMON_ROW_TPW") ? atoi(getenv("DEMON_ROW_TPW")) : 0;
    const int tpw = tpw_env > 0 ? tpw_env : 3;
    const int wgs = (ntiles + tpw - 1) / tpw;
    const dim3 grid((unsigned)(wgs < 1 ? 1 : wgs));
    const bool caffe = a.pad == taps / 2;
    if (taps == 9) return caffe ? launch_row_t<3, true>(a, grid, lds, stream) : launch_row_t<3, false>(a, grid, lds, stream);
    return caffe ? launch_row_t<2, true>(a, grid, lds, stream) : launch_row_t<2, false>(a, grid, lds, stream);
}

}  // namespace demon
