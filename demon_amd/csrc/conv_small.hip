// conv_small.hip -- the tiny contractions at the end of the heads, where a 32-row MFMA tile would be >= 87 % padding:
//   * conv_small_kernel: 3x3 stride-1 convs with Cout <= 4 (predict_flow*/conv2, predict_depthnormal2/conv2: 24 -> 4,
//     blocks_original.py:41-49, :263-271; predict_depth0/conv2: 16 -> 1, :511) as an LDS-tiled direct convolution on the
//     VALU: block = 8 x 128 output pixels, each thread 4 consecutive pixels x all output channels, input staged 8 channels
//     at a time with a 1-pixel halo, one ds_read_b128 + one ds_read_b64 per (channel, tap row).  Bandwidth bound.
//   * motion_tail_kernel: motion_fc2 (1024 -> 128, leaky relu), motion_fc3 (128 -> 7) and the split into rotation /
//     translation / scale (blocks_original.py:397-412) in one launch, one workgroup per sample.
// Both read the same packed weights Wp[k][Mpad] as the MFMA kernels (k = tap*Cin + ci).
#include "internal.h"

namespace demon {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

constexpr int SM_CH = 8, SM_TH = 8;

// PX: pixels per thread along x (4: tiles of 8 x 128 pixels; 2 (round 6): tiles of 8 x 64 -- the 48 x 64 heads of the blocks are 64 wide, so
// half of a 128-wide tile's threads computed pixels that do not exist: 26 us per launch for 11 MB)
template <int CO, int PX>
__global__ __launch_bounds__(256) void conv_small_kernel(SmallConvArgs a)
{
    constexpr int SM_TW = 32 * PX, SM_STRIDE = SM_TW + 8;  // 136 / 72 floats (cols 3 .. TW + 4 used): rows stay 16-byte aligned
    __shared__ __attribute__((aligned(16))) float tile[SM_CH][SM_TH + 2][SM_STRIDE];
    __shared__ __attribute__((aligned(16))) float wl[64 * 9 * CO];  // weights [ci][tap][co]: uniform (broadcast) LDS reads
    for (int e = threadIdx.x; e < a.Cin * 9 * CO; e += 256) {
        const int co = e % CO, t = e / CO;
        const int tap = t % 9, ci = t / 9;
        wl[e] = co < a.Cout ? a.wp[(long)(tap * a.Cin + ci) * a.Mpad + co] : 0.0f;
    }
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * SM_TW, y0 = blockIdx.y * SM_TH;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int hw = a.H * a.W;
    const float *__restrict__ inp = a.in + (long)n * a.in_n_stride;
    float acc[PX][CO];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[p][co] = 0.0f;

    for (int c0 = 0; c0 < a.Cin; c0 += SM_CH) {
        const int nc = min(SM_CH, a.Cin - c0);
        __syncthreads();
        // stage [nc][10 rows][130 columns x0-1 .. x0+128], zeros outside the image; LDS column = (gx - x0) + 4 so that a
        // thread's own four pixels start on a 16-byte boundary.  Index math is shifts only: two rows per pass for the
        // 128 interior columns, then one pass for the two halo columns of every (channel, row).
        {
            const bool vec = (a.W & 3) == 0 && (hw & 3) == 0 && (a.in_n_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;
            if (vec) {
                // 16 bytes per lane: 32 (16) lanes cover the interior columns of one (channel, row), 8 (16) of them per pass
                constexpr int LPR = SM_TW / 4, RPP = 256 / LPR;   // lanes per row, rows per pass
                const int sub = threadIdx.x / LPR, col = (threadIdx.x % LPR) * 4;
                const int gx = x0 + col;
                const bool xin = gx < a.W;  // W % 4 == 0: a group of four columns is inside or outside as a whole
                for (int rr = 0; rr < nc * (SM_TH + 2); rr += RPP) {
                    const int q = rr + sub;
                    if (q < nc * (SM_TH + 2)) {
                        const int c = q / (SM_TH + 2), row = q - c * (SM_TH + 2);
                        const int gy = y0 - 1 + row;
                        floatx4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (xin && gy >= 0 && gy < a.H) v = *reinterpret_cast<const floatx4 *>(inp + (long)(c0 + c) * hw + gy * a.W + gx);
                        *reinterpret_cast<floatx4 *>(&tile[c][row][col + 4]) = v;
                    }
                }
            } else {
            constexpr int RPP = 256 / SM_TW;   // rows per pass
            const int half = threadIdx.x / SM_TW, col = threadIdx.x % SM_TW;
            const int gx = x0 + col;
            const bool xin = gx < a.W;
            for (int rr = 0; rr < nc * (SM_TH + 2); rr += RPP) {
                const int q = rr + half;  // (channel, row) pair index
                if (q < nc * (SM_TH + 2)) {
                    const int c = q / (SM_TH + 2), row = q - c * (SM_TH + 2);  // division by the constant 10
                    const int gy = y0 - 1 + row;
                    float v = 0.0f;
                    if (xin && gy >= 0 && gy < a.H) v = inp[(long)(c0 + c) * hw + gy * a.W + gx];
                    tile[c][row][col + 4] = v;
                }
            }
            }
            const int q = threadIdx.x >> 1, side = threadIdx.x & 1;
            if (q < nc * (SM_TH + 2)) {
                const int c = q / (SM_TH + 2), row = q - c * (SM_TH + 2);
                const int gy = y0 - 1 + row, gxh = side ? x0 + SM_TW : x0 - 1;
                float v = 0.0f;
                if (gxh >= 0 && gxh < a.W && gy >= 0 && gy < a.H) v = inp[(long)(c0 + c) * hw + gy * a.W + gxh];
                tile[c][row][side ? SM_TW + 4 : 3] = v;
            }
        }
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                // values x-1 .. x+PX of row ty+dy: LDS columns PX*tx+3 .. PX*tx+4+PX
                const float *r = &tile[c][ty + dy][PX * tx];
                float v[PX + 2];
                v[0] = r[3];
                if constexpr (PX == 4) {
                    const floatx4 v4 = *reinterpret_cast<const floatx4 *>(r + 4);
                    v[1] = v4[0]; v[2] = v4[1]; v[3] = v4[2]; v[4] = v4[3];
                } else {
                    const floatx2 v2 = *reinterpret_cast<const floatx2 *>(r + 4);
                    v[1] = v2[0]; v[2] = v2[1];
                }
                v[PX + 1] = r[4 + PX];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float *w = &wl[((c0 + c) * 9 + dy * 3 + dx) * CO];  // wave uniform address: LDS broadcast
#pragma unroll
                    for (int co = 0; co < CO; ++co) {
                        const float wv = w[co];
#pragma unroll
                        for (int p = 0; p < PX; ++p) acc[p][co] = fmaf(v[p + dx], wv, acc[p][co]);
                    }
                }
            }
        }
    }
    const int x = x0 + PX * tx, y = y0 + ty;
    if (y >= a.H || x >= a.W) return;
    const float sc = a.scale ? a.scale[n] : 1.0f;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        if (co >= a.Cout) break;
        const float b = a.bias[co];
        float o[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            float v = acc[p][co] + b;
            if (a.act) v = v >= 0.0f ? v : 0.1f * v;
            if (co == 0) v *= sc;
            o[p] = v;
        }
        float *__restrict__ dst = a.out + (long)n * a.out_n_stride + (long)co * hw + y * a.W + x;
        if (x + PX - 1 < a.W && (a.W & 3) == 0) {
            if constexpr (PX == 4) *reinterpret_cast<floatx4 *>(dst) = floatx4{o[0], o[1], o[2], o[3]};
            else *reinterpret_cast<floatx2 *>(dst) = floatx2{o[0], o[1]};
        } else {
            for (int p = 0; p < PX && x + p < a.W; ++p) dst[p] = o[p];
        }
    }
}

bool conv_small_applies(int kh, int kw, int sh, int sw, int Cin, int Cout) { return kh == 3 && kw == 3 && sh == 1 && sw == 1 && Cout <= 4 && Cin <= 64; }

void launch_conv_small(const SmallConvArgs &a, int N, hipStream_t s)
{
    // maps up to 64 (or 64 + 64 k, k odd ...: whenever the last 128-wide tile would be at most half full) wide: 64-wide tiles
    const bool narrow = (a.W % 128) != 0 && (a.W % 128) <= 64;
    const int tw = narrow ? 64 : 128;
    dim3 grid((a.W + tw - 1) / tw, (a.H + SM_TH - 1) / SM_TH, N);
    if (narrow) {
        if (a.Cout <= 1) hipLaunchKernelGGL((conv_small_kernel<1, 2>), grid, dim3(256), 0, s, a);
        else if (a.Cout <= 2) hipLaunchKernelGGL((conv_small_kernel<2, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_small_kernel<4, 2>), grid, dim3(256), 0, s, a);
    } else {
        if (a.Cout <= 1) hipLaunchKernelGGL((conv_small_kernel<1, 4>), grid, dim3(256), 0, s, a);
        else if (a.Cout <= 2) hipLaunchKernelGGL((conv_small_kernel<2, 4>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_small_kernel<4, 4>), grid, dim3(256), 0, s, a);
    }
}

// one workgroup (256 threads) per sample: h2 = lrelu(W2^T x + b2) (128), m = W3^T h2 + b3 (7) -> rotation, translation, scale.
// fc2: thread (g = t / 32, j4 = t % 32) accumulates outputs 4*j4 .. 4*j4+3 over the K slice g (16-byte weight loads, a
// wavefront reads two full 512-byte weight rows per step), then the 8 slices are summed through LDS.
__global__ __launch_bounds__(256) void motion_tail_kernel(const float *__restrict__ x, const float *__restrict__ w2,
                                                          const float *__restrict__ b2, const float *__restrict__ w3,
                                                          const float *__restrict__ b3, float *__restrict__ motion,
                                                          float *__restrict__ rot, float *__restrict__ trans,
                                                          float *__restrict__ scale, int K2, int M2pad, int M3pad)
{
    __shared__ float xs[1024];
    __shared__ __attribute__((aligned(16))) float part[8][128];
    __shared__ float h2[128];
    const int n = blockIdx.x, t = threadIdx.x;
    for (int k = t; k < K2; k += 256) xs[k] = x[(long)n * K2 + k];
    __syncthreads();
    const int g = t >> 5, j4 = t & 31;
    const int ks = K2 / 8, kb = g * ks;
    floatx4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
    for (int k = kb; k < kb + ks; ++k) {
        const floatx4 w = *reinterpret_cast<const floatx4 *>(w2 + (long)k * M2pad + 4 * j4);
        const float xv = xs[k];
        acc[0] = fmaf(xv, w[0], acc[0]); acc[1] = fmaf(xv, w[1], acc[1]);
        acc[2] = fmaf(xv, w[2], acc[2]); acc[3] = fmaf(xv, w[3], acc[3]);
    }
    *reinterpret_cast<floatx4 *>(&part[g][4 * j4]) = acc;
    __syncthreads();
    if (t < 128) {
        float v = b2[t];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += part[q][t];
        h2[t] = v >= 0.0f ? v : 0.1f * v;
    }
    __syncthreads();
    if (t < 7) {
        float v = b3[t];
        for (int k = 0; k < 128; ++k) v = fmaf(h2[k], w3[(long)k * M3pad + t], v);
        motion[n * 7 + t] = v;
        if (t < 3) rot[n * 3 + t] = v;
        else if (t < 6) trans[n * 3 + t - 3] = v;
        else scale[n] = v;
    }
}

void launch_motion_tail(const float *x, const float *w2, const float *b2, const float *w3, const float *b3, float *motion,
                        float *rot, float *trans, float *scale, int N, int K2, int M2pad, int M3pad, hipStream_t s)
{
    hipLaunchKernelGGL(motion_tail_kernel, dim3(N), dim3(256), 0, s, x, w2, b2, w3, b3, motion, rot, trans, scale, K2, M2pad,
                       M3pad);
}

}  // namespace demon
