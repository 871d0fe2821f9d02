// dense_stream.hip -- the big fully connected layers at small batch: `dense5` of the v2 blocks (v2/blocks.py:197-213, :395-411: a
// square layer over the first 96 channels of conv5_1, 4608 x 4608 at 256 x 192 = 85 MB of weights) and `motion_fc1`
// (blocks_original.py:390-396, 6144 x 1024 = 25 MB).  With a batch of 32 every weight is used for 32 multiply-adds: the layer is a
// weight STREAM (HBM bound: 85 MB / 8 TB/s = 10.6 us against 8.6 us of MFMA time at peak), so the kernel is built around the stream:
//
//   * weights go global memory -> registers -> MFMA A operand, never through LDS: a lane loads 16 bytes = 4 consecutive output
//     units of one weight row, and the 4 values feed 4 MFMAs (32x32x2) whose row r stands for unit 4 r + i -- the permutation of
//     the output rows is undone at the store.  The weights are re-blocked once per weight update (dense_repack_kernel) to
//     [128-unit column block][K][128]: what a wave streams is one contiguous run of memory, 1 KB per load instruction (two
//     consecutive rows), 16 KB per chunk -- whole DRAM pages and TLB entries instead of 512-byte pieces 18 KB apart.
//   * a wave owns a contiguous run of weight rows (K) of a 128-unit column block and walks it in chunks of 4 groups of 8 rows:
//     16 loads of 16 bytes in flight per lane for the next chunk while the 64 MFMAs of the current one run (16 KB per wave,
//     8 waves per CU: well above the bytes in flight that 8 TB/s need), no barrier on the weight path.
//   * the activations (32 x K floats, L2 resident) are the MFMA B operand: the workgroup transposes its rows into LDS ([k][n],
//     pitch 33) one chunk ahead, double buffered, one barrier per chunk (64 MFMAs per wave).
//   * the 4 waves of a workgroup hold partial sums over different K ranges of the same outputs: combined through LDS in wave
//     order; K slices across workgroups (grid.y) go to the split-K workspace [slice][Mpad][N] and dense_reduce_kernel adds them
//     in slice order -- so results do not depend on scheduling.
#include "internal.h"

namespace demon {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int DS_NT = 256, DS_BM = 128, DS_BN = 32, DS_XP = 33;
constexpr int DS_CHUNK_ROWS = 128;                    // rows of x staged per chunk: 4 waves x 4 groups x 8 rows
constexpr int DS_XBUF = DS_CHUNK_ROWS * DS_XP;        // floats of one x buffer
constexpr int DS_LDS_FLOATS = 4 * 3 * 16 * 64;        // epilogue exchange (48 KB) >= the two x buffers (33 KB)

// NT: non-temporal policy on the weight loads (every byte is read once per launch); MODE: 0 product, 1 / 2 diagnostic builds of the
// loop (loads without MFMAs / MFMAs without loads) for tools/dense_probe.py
template <bool NT, int MODE>
__global__ __launch_bounds__(DS_NT, 2) void dense_stream_kernel(DenseArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[DS_LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * DS_BM, zs = blockIdx.y, n0 = blockIdx.z * DS_BN;
    constexpr int OOB = 0x7ffffff0;

    // K range of this workgroup (in groups of 8 weight rows), split evenly over its 4 waves
    const int G8 = a.K >> 3;
    const int g_begin = (int)((long)zs * G8 / a.ksplit), g_end = (int)((long)(zs + 1) * G8 / a.ksplit);
    const int ngroups = g_end - g_begin;
    int seg_begin[4], seg_cnt[4];   // per wave: first group (relative to g_begin) and number of groups
    int max_cnt = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        seg_begin[w] = w * ngroups / 4;
        seg_cnt[w] = (w + 1) * ngroups / 4 - seg_begin[w];
        max_cnt = max(max_cnt, seg_cnt[w]);
    }
    const int nchunks = (max_cnt + 3) >> 2;
    const int my_begin = wave == 0 ? seg_begin[0] : (wave == 1 ? seg_begin[1] : (wave == 2 ? seg_begin[2] : seg_begin[3]));
    const int my_cnt = wave == 0 ? seg_cnt[0] : (wave == 1 ? seg_cnt[1] : (wave == 2 ? seg_cnt[2] : seg_cnt[3]));

    // ---- weights: lane (r = l31, h = lhi) of MFMA step t of group g reads row 8 g + 2 t + h, units 4 r .. 4 r + 3 of the block
    const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wd + ((long)blockIdx.x * a.K + (long)g_begin * 8) * DS_BM), 0, rsrc_bytes(a.K * a.Mpad - ((int)blockIdx.x * a.K + g_begin * 8) * DS_BM), 0x00020000);   // [Mpad / 128][K][128]
    const int a_voff = 4 * (lhi * DS_BM + 4 * l31);
    constexpr int row_bytes = 4 * DS_BM;
    // ---- activations: thread (n = tid / 8, q = tid % 8) loads x[n][32 p + 4 q .. + 3] of the 32-row segment p (= wave p's groups)
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x + (long)n0 * a.x_n_stride + (long)g_begin * 8), 0, rsrc_bytes((a.N - 1 - n0) * (int)a.x_n_stride + a.K - g_begin * 8), 0x00020000);
    const int xn = tid >> 3, xq = tid & 7;
    const int x_voff = (n0 + xn < a.N) ? 4 * (xn * (int)a.x_n_stride + 4 * xq) : OOB;
    const int xs_store = (4 * xq) * DS_XP + xn;               // + (32 p + e) * DS_XP
    const int xs_read = (32 * wave + lhi) * DS_XP + l31;      // + (8 j + 2 t) * DS_XP

    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;

    floatx4 bufA[16], bufB[16], xr[4];
    auto load_x = [&](int c) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // group of this thread's 4 rows inside segment p: 4 c + q / 2; rows of groups the segment does not have read as zero
            const bool ok = 4 * c + (xq >> 1) < seg_cnt[p];
            xr[p] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? x_voff : OOB, 4 * 8 * (seg_begin[p] + 4 * c), 0));
        }
    };
    auto store_x = [&](int buf) {
        float *xs = smem + buf * DS_XBUF + xs_store;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[(32 * p + e) * DS_XP] = xr[p][e];
    };
    auto load_a = [&](floatx4 (&buf)[16], int c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = 4 * c + j;
            const int voff = g < my_cnt ? a_voff : OOB;   // (uniform) groups past the end: no memory traffic, zeros
            const int sbase = (my_begin + g) * 8 * row_bytes;
#pragma unroll
            for (int t = 0; t < 4; ++t) buf[j * 4 + t] = MODE == 2 ? floatx4{1.0f, 1.0f, 1.0f, 1.0f} : __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, voff, sbase + 2 * t * row_bytes, NT ? 2 : 0));
        }
    };
    auto compute = [&](const floatx4 (&buf)[16], int xbuf) {
        const float *xs = smem + xbuf * DS_XBUF + xs_read;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b = xs[(8 * j + 2 * t) * DS_XP];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (MODE == 1) acc[i][(j * 4 + t) & 15] += buf[j * 4 + t][i] * b;
                    else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[j * 4 + t][i], b, acc[i], 0, 0, 0);
                }
            }
    };

    if (nchunks > 0) {
        load_x(0);
        load_a(bufA, 0);
        store_x(0);
        __syncthreads();
        int c = 0;
        for (; c + 2 <= nchunks; c += 2) {
            load_x(c + 1);
            load_a(bufB, c + 1);
            compute(bufA, 0);
            store_x(1);
            __syncthreads();
            load_x(c + 2);          // past the last chunk: every offset is out of range (zeros, no traffic)
            load_a(bufA, c + 2);
            compute(bufB, 1);
            store_x(0);
            __syncthreads();
        }
        if (c < nchunks) compute(bufA, 0);
    }

    // ---- combine the 4 waves: wave w finishes accumulator set i = w (units m0 + 4 row + w); the other sets go through LDS
    __syncthreads();
    float *red = smem;   // [writer wave u][slot of set i != u][16][64]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i == wave) continue;
        const int slot = i < wave ? i : i - 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) red[((wave * 3 + slot) * 16 + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    const int n = n0 + l31;
    auto finish = [&](const floatx16 &own, int w) {   // w = this wave (compile-time in each call)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {             // wave order: the sum does not depend on which wave does it
                const int slot = w < u ? w : w - 1;
                v += u == w ? own[j] : red[((u * 3 + slot) * 16 + j) * 64 + lane];
            }
            const int row = 8 * (j >> 2) + 4 * lhi + (j & 3);
            const int m = m0 + 4 * row + w;
            if (n >= a.N) continue;
            if (a.ksplit > 1) {
                a.ws[((long)zs * a.Mpad + m) * a.N + n] = v;
            } else if (m < a.Cout) {
                v += a.bias[m];
                if (a.act) v = v >= 0.0f ? v : 0.1f * v;
                a.out[(long)n * a.out_n_stride + m] = v;
            }
        }
    };
    if (wave == 0) finish(acc[0], 0);
    else if (wave == 1) finish(acc[1], 1);
    else if (wave == 2) finish(acc[2], 2);
    else finish(acc[3], 3);
}

// adds the K slices in slice order, bias, leaky relu; a workgroup turns a [16 units][32 samples] tile of the workspace (read as it
// lies) into 32 runs of 16 consecutive units of the output.  Four slices per round, so that a thread has 8 loads in flight: the
// launch is a chain of memory latencies (ksplit / 4 of them), not a stream.  grid: (Mpad / 16, ceil(N / 32))
__global__ __launch_bounds__(256) void dense_reduce_kernel(DenseArgs a)
{
    __shared__ float tile[16 * 33];
    const int tid = threadIdx.x, m0 = blockIdx.x * 16, n0 = blockIdx.y * 32;
    const int nn = tid & 31, mm = tid >> 5;   // this thread: units m0 + mm and m0 + mm + 8
    const long stride = (long)a.Mpad * a.N;
    float v0 = 0.0f, v1 = 0.0f;
    if (n0 + nn < a.N) {
        const float *__restrict__ p0 = a.ws + (long)(m0 + mm) * a.N + n0 + nn;
        const float *__restrict__ p1 = p0 + 8l * a.N;
        int z = 0;
        for (; z + 4 <= a.ksplit; z += 4) {
            const float a0 = p0[z * stride], a1 = p0[(z + 1) * stride], a2 = p0[(z + 2) * stride], a3 = p0[(z + 3) * stride];
            const float b0 = p1[z * stride], b1 = p1[(z + 1) * stride], b2 = p1[(z + 2) * stride], b3 = p1[(z + 3) * stride];
            v0 = (((v0 + a0) + a1) + a2) + a3;
            v1 = (((v1 + b0) + b1) + b2) + b3;
        }
        for (; z < a.ksplit; ++z) { v0 += p0[z * stride]; v1 += p1[z * stride]; }
        v0 += a.bias[m0 + mm];       // (padded to Mpad)
        v1 += a.bias[m0 + mm + 8];
        if (a.act) { v0 = v0 >= 0.0f ? v0 : 0.1f * v0; v1 = v1 >= 0.0f ? v1 : 0.1f * v1; }
    }
    tile[mm * 33 + nn] = v0;
    tile[(mm + 8) * 33 + nn] = v1;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i, on = e >> 4, om = e & 15;
        if (n0 + on < a.N && m0 + om < a.Cout) a.out[(long)(n0 + on) * a.out_n_stride + m0 + om] = tile[om * 33 + on];
    }
}

// packed weights [K][Mpad] -> [Mpad / 128][K][128]; a thread moves 16 bytes
__global__ __launch_bounds__(256) void dense_repack_kernel(float *__restrict__ wd, const float *__restrict__ wp, int K, int Mpad)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;   // float4 index in the destination
    const long total = (long)K * Mpad / 4;
    if (e >= total) return;
    const int c4 = (int)(e & 31);
    const long rk = e >> 5;
    const int k = (int)(rk % K), mb = (int)(rk / K);
    reinterpret_cast<floatx4 *>(wd)[e] = *reinterpret_cast<const floatx4 *>(wp + (long)k * Mpad + mb * DS_BM + c4 * 4);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
void launch_dense_repack(float *wd, const float *wp, int K, int Mpad, hipStream_t s)
{
    const long total = (long)K * Mpad / 4;
    hipLaunchKernelGGL(dense_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wd, wp, K, Mpad);
}

bool dense_stream_geometry_ok(int K, int Mpad, int ksplit)
{
    if (K < 64 || (K & 7) || (Mpad % DS_BM) || ksplit < 1 || ksplit > K / 32) return false;
    const long per_wg_rows = ((long)K + ksplit - 1) / ksplit + 8;
    return per_wg_rows * DS_BM * 4 < 0x3f000000l;   // 32-bit byte offsets inside a workgroup's slab of weights
}

long dense_stream_workgroups(const DenseArgs &a) { return (long)(a.Mpad / DS_BM) * a.ksplit * ((a.N + DS_BN - 1) / DS_BN); }

void launch_dense_stream(const DenseArgs &a, int variant, hipStream_t stream)
{
    dim3 grid((unsigned)(a.Mpad / DS_BM), (unsigned)a.ksplit, (unsigned)((a.N + DS_BN - 1) / DS_BN));
    static const int mode = getenv("DEMON_DENSE_MODE") ? atoi(getenv("DEMON_DENSE_MODE")) : 0;   // tools/dense_probe.py
    if (mode == 1) hipLaunchKernelGGL((dense_stream_kernel<false, 1>), grid, dim3(DS_NT), 0, stream, a);
    else if (mode == 2) hipLaunchKernelGGL((dense_stream_kernel<false, 2>), grid, dim3(DS_NT), 0, stream, a);
    else if (variant == 1) hipLaunchKernelGGL((dense_stream_kernel<true, 0>), grid, dim3(DS_NT), 0, stream, a);
    else hipLaunchKernelGGL((dense_stream_kernel<false, 0>), grid, dim3(DS_NT), 0, stream, a);
    if (a.ksplit > 1) {
        // demon_profile_full times the reduce launch on its own (like conv_splitk_reduce)
        if (g_reduce_mark && hipEventRecord(g_reduce_mark, stream) == hipSuccess) g_reduce_marked = true;
        dim3 rgrid((unsigned)(a.Mpad / 16), (unsigned)((a.N + 31) / 32));
        hipLaunchKernelGGL(dense_reduce_kernel, rgrid, dim3(256), 0, stream, a);
    }
}

}  // namespace demon
