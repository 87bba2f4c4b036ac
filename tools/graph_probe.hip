// tools/graph_probe.hip -- a chain of T dependent small kernels (one per timestep): plain launches vs one hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -o tools/graph_probe tools/graph_probe.hip && tools/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void step(const float *__restrict__ in, float *__restrict__ out, int n, int t, int spin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = i < n ? in[i] : 0.f;
    for (int k = 0; k < spin; ++k) v = v * 1.0001f + 0.5f;
    if (i < n) out[i] = v + (float)t;
}

int main() {
    const int n = 1440 * 64, T = 500;
    float *a, *b;
    CHECK(hipMalloc(&a, n * sizeof(float)));
    CHECK(hipMalloc(&b, n * sizeof(float)));
    CHECK(hipMemset(a, 0, n * sizeof(float)));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int spin : {0, 200, 1000}) {
        auto chain = [&]() {
            for (int t = 1; t < T; ++t)
                hipLaunchKernelGGL(step, dim3(360), dim3(256), 0, s, (t & 1) ? a : b, (t & 1) ? b : a, n, t, spin);
        };
        chain();
        CHECK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 5; ++r) chain();
        CHECK(hipStreamSynchronize(s));
        const double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5 / (T - 1);
        hipGraph_t graph;
        hipGraphExec_t exec;
        auto c0 = std::chrono::steady_clock::now();
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        chain();
        CHECK(hipStreamEndCapture(s, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        const double build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
        CHECK(hipGraphLaunch(exec, s));
        CHECK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 5; ++r) CHECK(hipGraphLaunch(exec, s));
        CHECK(hipStreamSynchronize(s));
        const double graphed = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5 / (T - 1);
        printf("spin %4d: plain launches %.2f us per kernel; graph %.2f us per kernel (capture + instantiate %.2f ms)\n", spin, plain,
               graphed, build);
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
    }
    return 0;
}
