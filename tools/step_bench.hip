// tools/step_bench.hip -- times the dense step kernel alone (with optional ablations) on the
// headline shape.  Build (one binary per ablation):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
//         -DDENSE_ABLATE=N -Itorbi_amd/csrc -o tools/step_bench_N tools/step_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dense_forward.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int BL, int JL>
void run(int B, int T, int S, int reps) {
    dense::Plan pl = dense::make_plan(B, S, 256, BL);
    if (pl.JL != JL) { printf("plan JL %d != %d, skip\n", pl.JL, JL); return; }
    const size_t panel = (size_t)pl.n_bt * pl.Kp * pl.BT, trp = (size_t)pl.n_jt * pl.Kp * pl.W;
    float *p0, *p1, *tr, *hist, *obs; int *frames;
    CHECK(hipMalloc(&p0, panel * 4)); CHECK(hipMalloc(&p1, panel * 4)); CHECK(hipMalloc(&tr, trp * 4));
    CHECK(hipMalloc(&hist, (size_t)B * T * S * 4)); CHECK(hipMalloc(&obs, (size_t)B * T * S * 4));
    CHECK(hipMalloc(&frames, B * 4));
    std::vector<float> h(panel);
    for (size_t i = 0; i < panel; ++i) h[i] = -(float)(rand() % 16000) / 1000.f;
    CHECK(hipMemcpy(p0, h.data(), panel * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(p1, h.data(), panel * 4, hipMemcpyHostToDevice));
    std::vector<float> ht(trp);
    for (size_t i = 0; i < trp; ++i) ht[i] = -(float)(rand() % 16000) / 1000.f;
    CHECK(hipMemcpy(tr, ht.data(), trp * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(obs, 0, (size_t)B * T * S * 4));
    std::vector<int> hf(B, T);
    CHECK(hipMemcpy(frames, hf.data(), B * 4, hipMemcpyHostToDevice));
    const size_t lds = dense::lds_bytes<BL, JL>();
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&dense::step_dense_kernel<BL, JL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int ntiles = pl.n_bt * pl.n_jt, grid = 8 * ((ntiles + 7) / 8);
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto go = [&](int n) {
        for (int t = 1; t <= n; ++t)
            hipLaunchKernelGGL((dense::step_dense_kernel<BL, JL>), dim3(grid), dim3(512), lds, 0, obs, frames, tr,
                               (t & 1) ? p0 : p1, (t & 1) ? p1 : p0, hist, B, T, S, 1 + (t % (T - 1)), pl.n_bt, pl.n_jt,
                               pl.JT, pl.KS, pl.Kp, pl.RB);
    };
    go(20); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(a)); go(reps); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("ablate=%d BL=%d JL=%d grid=%d lds=%zu: %.2f us/step  (%.2f Tcell/s useful)\n", DENSE_ABLATE, BL, JL, grid, lds,
           best * 1e3 / reps, (double)B * S * S / (best * 1e-3 / reps) / 1e12);
    hipFree(p0); hipFree(p1); hipFree(tr); hipFree(hist); hipFree(obs); hipFree(frames);
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, T = 8, S = argc > 2 ? atoi(argv[2]) : 1440;
    run<8, 6>(B, T, S, 200);
    run<4, 6>(B, T, S, 200);
    return 0;
}
