// conv_thin.hip -- the first layer of every block: conv1y, the 9 x 1 stride-(2,1) conv over the 6 channels of `image_pair`
// (blocks_original.py:141, :331 through helpers.py:105-153; v2/blocks.py the same with 24 outputs).  54 multiply-adds per output
// and channel: 2.7 GFLOP at batch 32 against 138 MB of input + output -- the layer is HBM bound (31 us at 4.5 TB/s, 18 us of MFMA
// time), and the general kernels, built around long K loops, spend their time in prologue and epilogue (conv_patch: 63 us).
//
// The whole reduction (K = Cin * taps <= 56) is 14 MFMA K-steps, so the WEIGHTS LIVE IN REGISTERS (28 per lane, loaded once) and
// there is no K loop over memory: a workgroup stages the input patch of an 8-row x 64-column output tile (6 x 23 x 64 floats,
// 35 KB, 16-byte coalesced loads) and each wave computes 2 output rows x 64 columns x 32 channels as 8 pixel tiles x 2 channel
// blocks of v_mfma_f32_16x16x4_f32.  k = ci * taps + tap; the B operand of (pixel tile, K-step) is one ds_read_b32 at
// [lane base + per-lane k offset] + immediate(tile).  The patch is stored with its columns permuted (x -> (x % 4) * 16 + x / 4)
// so that pixel tile xt holds columns 4 * lane + xt: the four tiles of a row give a lane 4 consecutive output pixels, stored as
// 16 bytes (256 contiguous bytes per 16 lanes).  Four workgroups per CU overlap one another's load, compute and store phases.
#include "internal.h"

namespace demon {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH_TY = 8, TH_TX = 64, TH_NT = 256;

template <int TAPS, int STRIDE, int NSTEPS>
__global__ __launch_bounds__(TH_NT, 4) void conv_thin_kernel(ThinArgs a)
{
    constexpr int PH = (TH_TY - 1) * STRIDE + TAPS;   // patch rows
    extern __shared__ __attribute__((aligned(16))) float patch[];   // [Cin][PH][64], columns permuted
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, 0, gridDim.x, 1, bx, by);
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = bx / per_img, trem = bx - n * per_img;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int y0 = ty * TH_TY, x0 = tx * TH_TX;
    const int y_org = y0 * STRIDE - a.pad;
    constexpr int OOB = 0x7ffffff0;

    // ---- stage the patch: a thread moves 16-byte pieces (row = ci * PH + pr, 4 columns)
    const auto irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in + (long)n * a.in_n_stride + x0), 0, rsrc_bytes(view_floats_left(a.N, n, a.in_n_stride, a.Cin, 0, (long)a.H * a.W, (long)a.H * a.W) - x0), 0x00020000);
    const int nrows = a.Cin * PH;
    constexpr int MAXP = (6 * PH * 16 + TH_NT - 1) / TH_NT;   // Cin <= 6 (K <= 56)
    floatx4 pv[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int e = tid + i * TH_NT, row = e >> 4, xq = e & 15;
        const int ci = row / PH, pr = row - ci * PH;
        const int gy = y_org + pr;
        const bool ok = row < nrows && (unsigned)gy < (unsigned)a.H && x0 + 4 * xq < a.W;
        pv[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, ok ? 4 * ((ci * a.H + gy) * a.W + 4 * xq) : OOB, 0, 0));
    }

    // ---- weights: lane (co = l15 of channel block cb, k = 4 s + lk); packed rows are tap * Cin + ci
    float wa[NSTEPS][2];
    int koff[NSTEPS];
    const int K = a.Cin * TAPS;
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
        const int k = 4 * s + lk;
        const bool kv = k < K;
        const int ci = kv ? k / TAPS : 0, tap = kv ? k - ci * TAPS : 0;
        koff[s] = (ci * PH + tap) * TH_TX;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) wa[s][cb] = kv ? a.wp[(long)(tap * a.Cin + ci) * a.Mpad + cb * 16 + l15] : 0.0f;
    }

#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int e = tid + i * TH_NT, row = e >> 4, xq = e & 15;
        if (row < nrows) {
#pragma unroll
            for (int q = 0; q < 4; ++q) patch[row * TH_TX + q * 16 + xq] = pv[i][q];
        }
    }
    __syncthreads();

    // ---- 14 K-steps x 8 pixel tiles x 2 channel blocks; tile t = (row rr of this wave, column group xt)
    floatx4 acc[8][2];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[t][cb] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    const int base = (2 * wave * STRIDE) * TH_TX + l15;
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
        const float *p = patch + base + koff[s];
        float b[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) b[t] = p[(t >> 2) * STRIDE * TH_TX + (t & 3) * 16];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][cb], b[t], acc[t][cb], 0, 0, 0);
    }

    // ---- bias, leaky relu, 16-byte stores: lane = 4 consecutive pixels of channel cb * 16 + 4 lk + r
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (long)n * a.out_n_stride + x0, 0, rsrc_bytes(view_floats_left(a.N, n, a.out_n_stride, a.Cout, 0, a.out_plane, (long)a.Ho * a.Wo) - x0), 0x00020000);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int y = y0 + 2 * wave + rr;
        const bool yv = y < a.Ho && x0 + 4 * l15 < a.Wo;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cb * 16 + 4 * lk + r;
                const float bias = a.bias[co];   // (padded to Mpad)
                floatx4 v;
#pragma unroll
                for (int xt = 0; xt < 4; ++xt) {
                    float o = acc[rr * 4 + xt][cb][r] + bias;
                    if (a.act) o = fmaxf(o, 0.1f * o);
                    v[xt] = o;
                }
                const int off = (yv && co < a.Cout) ? 4 * (co * (int)a.out_plane + y * a.Wo + 4 * l15) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, 0);
            }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
bool conv_thin_shape_ok(int kh, int kw, int sh, int sw, int ph, int pw, int Cin, int Mpad, int W, int Wo)
{
    return kh == 9 && kw == 1 && sh == 2 && sw == 1 && ph >= 0 && ph <= 4 && pw == 0 &&   // ph 4: caffe padding, 3: 'same' (v2)
           Cin >= 1 && Cin * 9 <= 56 && Mpad == 32 && W == Wo && (W & 3) == 0;
}

void launch_conv_thin(ThinArgs a, hipStream_t stream)
{
    a.tiles_y = (a.Ho + TH_TY - 1) / TH_TY;
    a.tiles_x = (a.Wo + TH_TX - 1) / TH_TX;
    constexpr int PH = (TH_TY - 1) * 2 + 9;
    const size_t lds = sizeof(float) * (size_t)a.Cin * PH * TH_TX;
    hipLaunchKernelGGL((conv_thin_kernel<9, 2, 14>), dim3((unsigned)(a.N * a.tiles_y * a.tiles_x)), dim3(TH_NT), lds, stream, a);
}

}  // namespace demon
