// launch_probe.hip -- what a dependent kernel boundary costs on MI355X: chains of 200 kernels on one stream (eager and as a
// hipGraph), each (a) empty, (b) writing 12.6 MB, (c) reading the 12.6 MB the previous one wrote and writing 12.6 MB.
//   hipcc --offload-arch=gfx950 -O3 -o launch_probe tools/probes/launch_probe.hip && ./launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_empty(float *, const float *, int) {}
__global__ void k_write(float *out, const float *, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = 1.0f;
}
__global__ void k_copy(float *out, const float *in, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i] + 1.0f;
}

template <typename K>
static void run(const char *name, K kernel, int grid, float *a, float *b, int n)
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int chain = 200;
    auto enqueue = [&]() {
        for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, n);
    };
    enqueue();
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    enqueue();
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms_eager = 0;
    hipEventElapsedTime(&ms_eager, e0, e1);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    enqueue();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms_graph = 0;
    hipEventElapsedTime(&ms_graph, e0, e1);
    printf("%-28s grid %5d: %6.2f us per kernel eager, %6.2f us in a hipGraph\n", name, grid, 1e3f * ms_eager / chain, 1e3f * ms_graph / chain);
}

int main()
{
    const int n = 128 * 24 * 32 * 32;  // 12.6 MB: one level-3 activation tensor at batch 32
    float *a, *b;
    hipMalloc(&a, sizeof(float) * n);
    hipMalloc(&b, sizeof(float) * n);
    hipMemset(a, 0, sizeof(float) * n);
    hipMemset(b, 0, sizeof(float) * n);
    for (int grid : {64, 768, 3072}) {
        run("empty", k_empty, grid, a, b, n);
        run("write 12.6 MB", k_write, grid, a, b, n);
        run("read + write 12.6 MB", k_copy, grid, a, b, n);
    }
    return 0;
}
