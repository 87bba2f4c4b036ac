// tools/ubench6.hip -- can the scalar path (s_load_dwordx16 -> SGPR operands) feed the add,add,max3
// stream?  Each wave streams its own slice of an L2-resident panel; per 32 scalars it runs 48 VALU
// (16 states x 2 prev-states: add, add, max3) on one per-lane posterior pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <int CONSUME>
__global__ __launch_bounds__(1024) void k(const float *__restrict__ tr, const float *__restrict__ pp, float *out,
                                          int floats_per_wave, int passes, int region_floats) {
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int base = (int)(((long long)wave * floats_per_wave) % region_floats);
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = -1e30f;
    float p0 = pp[threadIdx.x], p1 = pp[threadIdx.x + 1024];
    for (int ps = 0; ps < passes; ++ps) {
        for (int i = 0; i < floats_per_wave; i += 32) {
            const float *q = tr + base + i;          // uniform address -> s_load_dwordx16 x2
            if (CONSUME) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = max3(acc[j], q[j] + p0, q[16 + j] + p1);
            } else {
                acc[0] = max3(acc[0], q[0] + p0, q[16] + p1);
            }
            asm volatile("" : "+v"(p0), "+v"(p1));
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F f) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f(); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float *tr, *pp, *out;
    const size_t NF = (16 << 20) + (1 << 16);
    CHECK(hipMalloc(&tr, NF * 4)); CHECK(hipMalloc(&pp, 1 << 16)); CHECK(hipMalloc(&out, 1 << 24));
    CHECK(hipMemset(tr, 0, NF * 4)); CHECK(hipMemset(pp, 0, 1 << 16));
    const int fpw = 8640;        // 34.5 KB per wave per pass (one wave's share of a 276 KB panel)
    const int passes = 60;
    for (int region_kb : {2048, 65536}) {            // per-XCD working set 2 MB (L2 resident) / 64 MB
        for (int threads : {512, 1024}) {
            for (int wg_per_cu : {1, 2}) {
                const int grid = 256 * wg_per_cu;
                const double scalars = (double)grid * (threads / 64) * fpw * passes;
                float ms = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(threads), 0, 0, tr, pp, out, fpw, passes, region_kb * 256); });
                printf("region %5d KB, %2d waves/CU: %.3f ms  %.2f Tcell/s  %.2f TB/s scalar stream  (%.1f B/clk/CU @2.1GHz)\n", region_kb,
                       threads / 64 * wg_per_cu, ms, scalars / ms / 1e9, scalars * 4 / ms / 1e9, scalars * 4 / (ms * 1e-3) / 256 / 2.1e9);
            }
        }
    }
    return 0;
}
