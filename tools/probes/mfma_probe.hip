// mfma_probe.hip -- what limits an fp32-MFMA K loop on gfx950?  Builds the conv kernel's inner structure up
// step by step (V0 pure MFMA ... V4 LDS tile + barrier + global loads) and prints TF/s for each.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/probes/mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int V, int WAVES_PER_SIMD_HINT>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ out, int nsteps)
{
    __shared__ __attribute__((aligned(16))) float As[2][16][128];
    __shared__ __attribute__((aligned(16))) float Bs[2][16][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fill LDS once
    for (int i = tid; i < 2 * 16 * 128; i += 256) { (&As[0][0][0])[i] = A[i]; (&Bs[0][0][0])[i] = B[i]; }
    __syncthreads();
    float av[8][2], bv[8][2];
    for (int kk = 0; kk < 8; ++kk) for (int i = 0; i < 2; ++i) { av[kk][i] = A[tid + kk * 2 + i]; bv[kk][i] = B[tid + kk * 2 + i + 7]; }
    floatx4 areg[2] = {0, 0};
    float breg[8] = {0};
    const float *ap = A + (size_t)blockIdx.x * 4096 + tid * 4;
    const float *bp = B + (size_t)blockIdx.x * 4096 + tid;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (V >= 1) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = 2 * kk + lhi;
#pragma unroll
                for (int i = 0; i < 2; ++i) av[kk][i] = As[buf][k][(wm * 2 + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[kk][j] = Bs[buf][k][(wn * 2 + j) * 32 + l31];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][i], bv[g][j], acc[i][j], 0, 0, 0);
            if (V == 4) {
                breg[g] = bp[(size_t)(s & 63) * 256 + g * 32768];
                if (g < 2) areg[g] = *reinterpret_cast<const floatx4 *>(ap + (size_t)(s & 63) * 1024 + g * 65536);
            }
            if (V == 5) {
                const int wave = tid >> 6;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bp + (size_t)(s & 63) * 256 + g * 32768),
                                                 (__attribute__((address_space(3))) void *)(&Bs[buf ^ 1][(tid >> 7) * 8 + g][(wave & 1) * 64]), 4, 0, 0);
                if (g < 2)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ap + (size_t)(s & 63) * 1024 + g * 65536),
                                                     (__attribute__((address_space(3))) void *)(&As[buf ^ 1][g * 8 + wave * 2][0]), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (V == 5) {
            // direct-to-LDS DMA: destination = wave-uniform base + lane * size
            __shared__ int dummy_guard;
            (void)dummy_guard;
        }
        if (V >= 3 && V != 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) Bs[buf ^ 1][(tid >> 7) * 8 + i][tid & 127] = breg[i] + (V >= 4 ? 0.f : (float)s);
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<floatx4 *>(&As[buf ^ 1][(tid >> 5) + i * 8][(tid & 31) * 4]) = areg[i];
        }
        if (V >= 2) __syncthreads();  // V5: the barrier's vmcnt(0) also drains the DMA
    }
    float sum = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = sum;
}

template <int V>
static void run(const float *A, const float *B, float *out, int wgs, int nsteps, const char *what)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<V, 3>), dim3(wgs), dim3(256), 0, 0, A, B, out, nsteps);
    hipEventRecord(e0);
    const int iters = 10;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<V, 3>), dim3(wgs), dim3(256), 0, 0, A, B, out, nsteps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double flops = (double)wgs * 4 * nsteps * 32 * 4096.0;
    printf("V%d %-46s wgs %5d  %8.3f ms  %7.1f TF/s\n", V, what, wgs, ms, flops / (ms * 1e-3) / 1e12);
}

int main()
{
    const size_t n = 64u << 20;
    float *A, *B, *out;
    hipMalloc(&A, n * 4); hipMalloc(&B, n * 4); hipMalloc(&out, 4096 * 256 * 4);
    std::vector<float> h(n);
    unsigned st = 1;
    for (auto &v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
    hipMemcpy(A, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int wgs : {768, 3072}) {
        run<0>(A, B, out, wgs, 72, "pure MFMA (operands in registers)");
        run<1>(A, B, out, wgs, 72, "+ LDS fragment reads every step");
        run<2>(A, B, out, wgs, 72, "+ barrier every step");
        run<3>(A, B, out, wgs, 72, "+ LDS tile writes every step");
        run<4>(A, B, out, wgs, 72, "+ global loads interleaved (8 dword + 2 dwordx4)");
        run<5>(A, B, out, wgs, 72, "same loads as direct-to-LDS DMA, no ds_write");
    }
    for (int wgs : {256, 512, 768, 1024, 1536})
        for (int ns : {18, 72, 288}) { char b[64]; snprintf(b, 64, "pure MFMA nsteps=%d", ns); run<0>(A, B, out, wgs, ns, b); }
    for (int ns : {18, 72, 288}) { char b[64]; snprintf(b, 64, "full (V4) nsteps=%d", ns); run<4>(A, B, out, 768, ns, b); }
    return 0;
}
