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

template <int BL, int JL, int NW, int KC, int MSL = 8>
void run(int B, int T, int S, int reps) {
    dense::Plan pl = dense::make_plan(B, S, 256, BL, NW);
    if (pl.NW != NW || pl.KC != KC) { printf("plan NW/KC %d/%d != %d/%d, skip\n", pl.NW, pl.KC, NW, KC); return; }
    if (getenv("NJT")) {           // experiment: force the number of state tiles
        pl.n_jt = atoi(getenv("NJT"));
        pl.JT = (S + pl.n_jt - 1) / pl.n_jt;
        pl.n_jt = (S + pl.JT - 1) / pl.JT;
    }
    if (getenv("RB")) pl.RB = atoi(getenv("RB"));
    if (pl.JL != JL) { printf("plan JL %d != %d, skip\n", pl.JL, JL); return; }
    const size_t panel = (size_t)pl.n_bt * pl.Kp * pl.BT, trp = (size_t)pl.n_jt * pl.Kp * pl.W;
    float *p0, *p1, *tr, *hist, *obs; int *frames; int *chunks;
    CHECK(hipMalloc(&p0, panel * 4)); CHECK(hipMalloc(&p1, panel * 4)); CHECK(hipMalloc(&tr, trp * 4));
    CHECK(hipMalloc(&hist, (size_t)B * T * S * 4)); CHECK(hipMalloc(&obs, (size_t)B * T * S * 4));
    CHECK(hipMalloc(&frames, B * 4));
    CHECK(hipMalloc(&chunks, (size_t)pl.n_jt * (pl.NCH + 1) * 4));
    std::vector<float> h(panel);
    for (size_t i = 0; i < panel; ++i) h[i] = -(float)(rand() % 16000) / 1000.f;
    CHECK(hipMemcpy(p0, h.data(), panel * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(p1, h.data(), panel * 4, hipMemcpyHostToDevice));
    std::vector<float> ht(trp);
    for (size_t i = 0; i < trp; ++i) ht[i] = -(float)(rand() % 16000) / 1000.f;
    CHECK(hipMemcpy(tr, ht.data(), trp * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(obs, 0, (size_t)B * T * S * 4));
    hipLaunchKernelGGL(dense::build_chunk_lists_kernel, dim3(pl.n_jt), dim3(256), sizeof(int) * (size_t)pl.NCH, 0, tr, chunks, S,
                       pl.JT, pl.W, pl.Kp, pl.NCH, pl.KC);
    std::vector<int> hf(B, T);
    CHECK(hipMemcpy(frames, hf.data(), B * 4, hipMemcpyHostToDevice));
    const size_t lds = dense::lds_bytes<BL, JL, NW, KC, MSL>();
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&dense::step_dense_kernel<BL, JL, NW, KC, MSL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int ntiles = pl.n_bt * pl.n_jt, grid = 8 * ((ntiles + 7) / 8);
    {
        int nb = 0;
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dense::step_dense_kernel<BL, JL, NW, KC, MSL>, 64 * NW, lds));
        hipFuncAttributes fa;
        CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&dense::step_dense_kernel<BL, JL, NW, KC, MSL>)));
        printf("   occupancy API: %d blocks/CU, numRegs %d, static LDS %zu, dyn LDS %zu\n", nb, fa.numRegs, fa.sharedSizeBytes, lds);
    }
    float *q0, *q1; CHECK(hipMalloc(&q0, panel * 4)); CHECK(hipMalloc(&q1, panel * 4));
    CHECK(hipMemcpy(q0, h.data(), panel * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(q1, h.data(), panel * 4, hipMemcpyHostToDevice));
    hipStream_t s0, s1; CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
    const bool two = getenv("TWO") != nullptr;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto go2 = [&](int n) {      // two independent step chains on two streams (two decodes in flight)
        for (int t = 1; t <= n; ++t) {
            hipLaunchKernelGGL((dense::step_dense_kernel<BL, JL, NW, KC, MSL>), dim3(grid), dim3(64 * NW), lds, s0, obs, frames, tr,
                               (t & 1) ? p0 : p1, (t & 1) ? p1 : p0, hist, chunks, B, T, S, 1 + (t % (T - 1)), pl.n_bt,
                               pl.n_jt, pl.JT, pl.Kp, pl.NCH, pl.RB);
            hipLaunchKernelGGL((dense::step_dense_kernel<BL, JL, NW, KC, MSL>), dim3(grid), dim3(64 * NW), lds, s1, obs, frames, tr,
                               (t & 1) ? q0 : q1, (t & 1) ? q1 : q0, hist, chunks, B, T, S, 1 + (t % (T - 1)), pl.n_bt,
                               pl.n_jt, pl.JT, pl.Kp, pl.NCH, pl.RB);
        }
    };
    if (two) {
        go2(20); CHECK(hipDeviceSynchronize());
        float best2 = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a, s0)); CHECK(hipStreamWaitEvent(s1, a, 0)); go2(reps);
            CHECK(hipEventRecord(b, s1)); CHECK(hipStreamWaitEvent(s0, b, 0)); CHECK(hipEventRecord(b, s0));
            CHECK(hipEventSynchronize(b)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best2) best2 = ms;
        }
        printf("TWO streams BL=%d JL=%d NW=%d KC=%d MSL=%d lds=%zu: %.2f us per step (each of the 2 chains advances one step per 2x that)\n",
               BL, JL, NW, KC, MSL, lds, best2 * 1e3 / (2 * reps));
    }
    auto go = [&](int n) {
        for (int t = 1; t <= n; ++t)
            hipLaunchKernelGGL((dense::step_dense_kernel<BL, JL, NW, KC, MSL>), dim3(grid), dim3(64 * NW), lds, 0, obs, frames, tr,
                               (t & 1) ? p0 : p1, (t & 1) ? p1 : p0, hist, chunks, B, T, S, 1 + (t % (T - 1)), pl.n_bt,
                               pl.n_jt, pl.JT, pl.Kp, pl.NCH, pl.RB);
    };
    go(20); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(a)); go(reps); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("ablate=%d BL=%d JL=%d NW=%d KC=%d grid=%d lds=%zu: %.2f us/step  (%.2f Tcell/s useful)\n", DENSE_ABLATE, BL, JL, NW, KC, grid, lds,
           best * 1e3 / reps, (double)B * S * S / (best * 1e-3 / reps) / 1e12);
#if DENSE_TIMING
    {
        go(1); CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> tb(4096 * 8);
        CHECK(hipMemcpyFromSymbol(tb.data(), HIP_SYMBOL(dense::timing_buf), tb.size() * 8));
        double acc[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long tmin = ~0ull, tmax = 0;
        const int nw = grid * NW;
        for (int w = 0; w < nw; ++w) {
            const unsigned long long *t = &tb[w * 8];
            acc[0] += (double)(t[1] - t[0]);   // setup
            acc[1] += (double)(t[5] - t[1]);   // first chunk landed
            acc[2] += (double)(t[2] - t[5]);   // contraction loop
            acc[3] += (double)(t[3] - t[2]);   // merge write + barriers
            acc[4] += (double)(t[4] - t[3]);   // finalize
            if (t[0] < tmin) tmin = t[0];
            if (t[4] > tmax) tmax = t[4];
        }
        {   // per-XCD span (clocks differ between XCDs): blocks b with b % 8 == 0
            unsigned long long lo = ~0ull, hi = 0, late = 0; int n = 0;
            for (int blk = 0; blk < grid; blk += 8) {
                const unsigned long long *t = &tb[(blk * NW) * 8];
                if (t[0] < lo) lo = t[0];
                if (t[4] > hi) hi = t[4];
            }
            for (int blk = 0; blk < grid; blk += 8) {
                const unsigned long long *t = &tb[(blk * NW) * 8];
                if (t[0] - lo > 5000) ++late;
                ++n;
            }
            printf("   XCD0: first start -> last end %llu ticks; %llu of %d blocks started > 5000 ticks after the first\n", hi - lo, late, n);
        }
        printf("   timing (s_memtime ticks, mean per wave): setup %.0f | first chunk %.0f | K loop %.0f | merge %.0f | finalize %.0f | first start -> last end %llu\n",
               acc[0] / nw, acc[1] / nw, acc[2] / nw, acc[3] / nw, acc[4] / nw, tmax - tmin);
    }
#endif
    (void)hipFree(p0); (void)hipFree(p1); (void)hipFree(tr); (void)hipFree(hist); (void)hipFree(obs); (void)hipFree(frames);
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, T = 8, S = argc > 2 ? atoi(argv[2]) : 1440;
    run<8, 6, 8, 12>(B, T, S, 200);
    run<8, 6, 16, 6>(B, T, S, 200);
    return 0;
}
