// chain_probe.hip -- how fast does ONE wave issue v_mfma_f32_32x32x2_f32 on gfx950 when every MFMA depends on the previous one
// (a wave that owns a single 32x32 accumulator block, the deep-layer tile of conv_mfma<128,32> / conv_stream<*,1>), with 2 / 4
// independent accumulators, with other waves on the SIMD, and with global loads issued between the MFMA groups?
//   hipcc --offload-arch=gfx950 -O3 -o chain_probe tools/probes/chain_probe.hip && ./chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// NACC accumulators round robin; LOADS: 0 none, 1 = 8 dword + 2 dwordx4 global loads per 8 MFMAs (consumed two groups later)
template <int NACC, int LOADS>
__global__ void chain(const float *__restrict__ src, float *__restrict__ out, unsigned long long *cyc, int groups)
{
    const int tid = threadIdx.x;
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = src[tid], b = src[tid + 64];
    float keep = 0.f;
    const float *p = src + (size_t)blockIdx.x * 65536 + tid;
    float ld[2][8];
    floatx4 l4[2][2];
    for (int i = 0; i < 8; ++i) { ld[0][i] = 0; ld[1][i] = 0; }
    l4[0][0] = l4[0][1] = l4[1][0] = l4[1][1] = floatx4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int g = 0; g < groups; ++g) {
        const int cur = g & 1;
        if (LOADS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ld[cur][i] = p[(size_t)((g * 8 + i) & 255) * 256];
            l4[cur][0] = *reinterpret_cast<const floatx4 *>(p + (size_t)(g & 63) * 1024 + 4 * tid - tid);
            l4[cur][1] = *reinterpret_cast<const floatx4 *>(p + (size_t)(g & 63) * 1024 + 512 + 4 * tid - tid);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float bb = LOADS ? (b + ld[cur ^ 1][k]) : b;
            acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[k % NACC], 0, 0, 0);
        }
        if (LOADS) keep += l4[cur ^ 1][0][0] + l4[cur ^ 1][1][3];
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = clock64();
    float s = keep;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
    if ((tid & 63) == 0 && blockIdx.x == 7) cyc[tid >> 6] = t1 - t0;
}

template <int NACC, int LOADS>
static void run(const float *src, float *out, unsigned long long *cyc, int blocks, int threads, const char *what)
{
    const int groups = 256;  // 2048 MFMAs per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<NACC, LOADS>), dim3(blocks), dim3(threads), 0, 0, src, out, cyc, groups);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<NACC, LOADS>), dim3(blocks), dim3(threads), 0, 0, src, out, cyc, groups);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double mfmas = (double)groups * 8;
    const double flops = (double)blocks * (threads / 64) * mfmas * 32 * 32 * 2 * 2;
    printf("%-44s blocks %4d x %4d thr | wave 0: %6.1f clk/MFMA (s_memtime) | %7.3f ms %6.1f TF/s\n", what, blocks, threads, h[0] / mfmas, ms,
           flops / (ms * 1e-3) / 1e12);
}

int main()
{
    float *src, *out; unsigned long long *cyc;
    hipMalloc(&src, (256u << 20)); hipMemset(src, 0, 256u << 20);
    hipMalloc(&out, 64u << 20); hipMalloc(&cyc, 256);
    for (int threads : {64, 256, 512, 768, 1024}) {   // 64: one wave per CU; 256: one per SIMD; 512 / 768 / 1024: 2 / 3 / 4 per SIMD
        run<1, 0>(src, out, cyc, 256, threads, "1 accumulator (dependent chain)");
        run<2, 0>(src, out, cyc, 256, threads, "2 accumulators");
        run<4, 0>(src, out, cyc, 256, threads, "4 accumulators");
    }
    for (int threads : {256, 768}) {
        run<1, 1>(src, out, cyc, 256, threads, "1 accumulator + 10 loads per 8 MFMAs");
        run<2, 1>(src, out, cyc, 256, threads, "2 accumulators + 10 loads per 8 MFMAs");
    }
    return 0;
}
