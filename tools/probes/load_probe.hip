// load_probe.hip -- cost of a few global loads per K-step inside an fp32-MFMA loop (3 waves/SIMD, 768 WGs).
//   NL  = dword loads per thread per step, SRC = 0: L2-resident 1 MB region, 1: streaming (every WG/step its own
//   64-lane rows from a 1 GB buffer), DIST = prefetch distance in steps (1 or 2), LDSW = also write them to LDS + barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NL, int SRC, int DIST, int LDSW>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ A, float *__restrict__ out, int nsteps)
{
    __shared__ float Ls[2][16][128];
    const int tid = threadIdx.x;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float av[8][2], bv[8][2];
    for (int kk = 0; kk < 8; ++kk) for (int i = 0; i < 2; ++i) { av[kk][i] = A[tid + kk * 2 + i]; bv[kk][i] = A[tid + kk * 2 + i + 7]; }
    float r0[NL > 0 ? NL : 1], r1[NL > 0 ? NL : 1];
    for (int i = 0; i < NL; ++i) { r0[i] = 0; r1[i] = 0; }
    const size_t wg_base = SRC ? (size_t)blockIdx.x * (size_t)nsteps * NL * 256 : 0;
    auto addr = [&](int s, int i) -> const float * {
        return SRC ? A + wg_base + ((size_t)s * NL + i) * 256 + tid : A + (((size_t)(s & 15) * NL + i) * 256 + tid + (blockIdx.x & 63) * 4096);
    };
    float sum = 0;
    for (int s = 0; s < nsteps; ++s) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][i], bv[g][j], acc[i][j], 0, 0, 0);
            if (g < 4) {
#pragma unroll
                for (int i = g * NL / 4; i < (g + 1) * NL / 4; ++i) {
                    if (DIST == 1) r0[i] = *addr(s + 1, i);
                    else if (s & 1) r1[i] = *addr(s + 2, i); else r0[i] = *addr(s + 2, i);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // consume the set that must have arrived by now
        if (LDSW) {
#pragma unroll
            for (int i = 0; i < NL; ++i) Ls[s & 1][(tid >> 7) * 8 + (i & 7)][tid & 127] = (DIST == 1 || (s & 1)) ? r0[i] : r1[i];
            __syncthreads();
            sum += Ls[s & 1][s & 15][tid & 127];
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) sum += (DIST == 1 || (s & 1)) ? r0[i] : r1[i];
        }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = sum;
}

template <int NL, int SRC, int DIST, int LDSW>
static void run(const float *A, float *out)
{
    const int wgs = 768, nsteps = 72;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<NL, SRC, DIST, LDSW>), dim3(wgs), dim3(256), 0, 0, A, out, nsteps);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<NL, SRC, DIST, LDSW>), dim3(wgs), dim3(256), 0, 0, A, out, nsteps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("NL=%2d src=%s dist=%d ldsw=%d   %7.3f ms  %6.1f TF/s\n", NL, SRC ? "stream" : "L2    ", DIST, LDSW, ms, (double)wgs * 4 * nsteps * 32 * 4096.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    const size_t n = 768ull * 72 * 16 * 256 + (1 << 20);
    float *A, *out;
    hipMalloc(&A, n * 4); hipMalloc(&out, 768 * 256 * 4);
    hipMemset(A, 0, n * 4);
    std::vector<float> h(1 << 20);
    unsigned st = 1;
    for (auto &v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 0, 1, 0>(A, out); run<0, 0, 1, 0>(A, out);
    run<2, 0, 1, 0>(A, out); run<4, 0, 1, 0>(A, out); run<8, 0, 1, 0>(A, out); run<16, 0, 1, 0>(A, out);
    run<2, 1, 1, 0>(A, out); run<4, 1, 1, 0>(A, out); run<8, 1, 1, 0>(A, out); run<16, 1, 1, 0>(A, out);
    run<4, 1, 2, 0>(A, out); run<8, 1, 2, 0>(A, out);
    run<4, 0, 1, 1>(A, out); run<4, 1, 1, 1>(A, out); run<8, 1, 1, 1>(A, out); run<4, 1, 2, 1>(A, out); run<8, 1, 2, 1>(A, out);
    return 0;
}
