// Dependent tiny kernels: per-kernel cost enqueued one by one on a stream against the same chain replayed from a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void k_small(float* p, int n, int rounds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float v = p[i];
        for (int r = 0; r < rounds; ++r) v = v * 1.0001f + 0.5f;
        p[i] = v;
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int n = 512 * 512;
    float* p; (void)hipMalloc(&p, n * sizeof(float)); (void)hipMemset(p, 0, n * sizeof(float));
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rounds : {1, 400}) {
        const int N = 2000;
        hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, p, n, rounds);
        (void)hipStreamSynchronize(s);
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, p, n, rounds);
        double t_enq = now() - t0;
        (void)hipStreamSynchronize(s);
        double t_stream = now() - t0;
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, p, n, rounds);
        (void)hipStreamEndCapture(s, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        t0 = now();
        for (int i = 0; i < N / 50; ++i) (void)hipGraphLaunch(ge, s);
        (void)hipStreamSynchronize(s);
        double t_graph = now() - t0;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, p, n, rounds);
        (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("kernel of %d rounds (alone: %.1f us by events): stream %.2f us per dependent launch (host enqueue %.2f us), graph of 50 nodes %.2f us per node\n",
               rounds, ms * 1e3, t_stream / N * 1e6, t_enq / N * 1e6, t_graph / N * 1e6);
    }
    return 0;
}
