// tools/ubench3.hip -- do LDS fragment reads overlap with the add,add,max3 stream?
// Variants: V = VALU only, L = LDS reads only, B = both (software-pipelined as in the step kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// MODE bit0: VALU, bit1: LDS reads.  PATTERN 0: step-kernel addresses (8 distinct P, 8 distinct T);
// 1: every lane its own 16 B (no broadcast)
template <int MODE, int PATTERN>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += blockDim.x) sm[i] = (float)i * 1e-3f;
    __syncthreads();
    const int bg = lane & 7, jg = lane >> 3;
    const float *lp = sm + (wave & 7) * 768;
    const float *lt = sm + 6144 + (wave & 7) * 128;
    float acc[8][6];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a][b] = -1e30f;
    float p0[8], p1[8], t0[6], t1[6];
#pragma unroll
    for (int a = 0; a < 8; ++a) { p0[a] = lane + a; p1[a] = lane - a; }
#pragma unroll
    for (int b = 0; b < 6; ++b) { t0[b] = b; t1[b] = -b; }
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        float q0[8], q1[8], u0[6], u1[6];
        if (MODE & 2) {
            const int ip = (it & 3) * 2;
            const int po = PATTERN == 0 ? 4 * bg : 4 * lane;
            const int to = PATTERN == 0 ? 4 * jg : 4 * lane;
            const float4 a = *reinterpret_cast<const float4 *>(&lp[ip * 64 + po]);
            const float4 b = *reinterpret_cast<const float4 *>(&lp[ip * 64 + 32 + po]);
            const float4 c = *reinterpret_cast<const float4 *>(&lp[(ip + 1) * 64 + po]);
            const float4 d = *reinterpret_cast<const float4 *>(&lp[(ip + 1) * 64 + 32 + po]);
            const float4 e = *reinterpret_cast<const float4 *>(&lt[ip * 8 + to]);
            const float4 f = *reinterpret_cast<const float4 *>(&lt[(ip + 1) * 8 + to]);
            const float2 g = *reinterpret_cast<const float2 *>(&lt[ip * 8 + 32 + 2 * jg]);
            const float2 h = *reinterpret_cast<const float2 *>(&lt[(ip + 1) * 8 + 32 + 2 * jg]);
            q0[0] = a.x; q0[1] = a.y; q0[2] = a.z; q0[3] = a.w; q0[4] = b.x; q0[5] = b.y; q0[6] = b.z; q0[7] = b.w;
            q1[0] = c.x; q1[1] = c.y; q1[2] = c.z; q1[3] = c.w; q1[4] = d.x; q1[5] = d.y; q1[6] = d.z; q1[7] = d.w;
            u0[0] = e.x; u0[1] = e.y; u0[2] = e.z; u0[3] = e.w; u0[4] = g.x; u0[5] = g.y;
            u1[0] = f.x; u1[1] = f.y; u1[2] = f.z; u1[3] = f.w; u1[4] = h.x; u1[5] = h.y;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 1) {
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) acc[a][b] = max3(acc[a][b], p0[a] + t0[b], p1[a] + t1[b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 2) {
            if (MODE & 1) {
#pragma unroll
                for (int a = 0; a < 8; ++a) { p0[a] = q0[a]; p1[a] = q1[a]; }
#pragma unroll
                for (int b = 0; b < 6; ++b) { t0[b] = u0[b]; t1[b] = u1[b]; }
            } else {
#pragma unroll
                for (int a = 0; a < 8; ++a) sink += q0[a] + q1[a];
#pragma unroll
                for (int b = 0; b < 6; ++b) sink += u0[b] + u1[b];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) sink += acc[a][b];
    out[blockIdx.x * blockDim.x + tid] = sink;
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

template <int MODE, int PATTERN>
void run(const char *name, float *out) {
    const int iters = 4000;
    for (int threads : {512, 1024}) {
        float ms = time_ms([&] { hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(256), dim3(threads), 65536, 0, out, iters); });
        const double per_iter_cyc = ms * 1e-3 * 2.0e9 / iters;
        printf("%-28s %2d waves/CU: %.3f ms  %.0f cycles/iter (@2.0GHz)  [iter = 96 cells/lane, 28 dwords LDS/lane]\n", name, threads / 64, ms, per_iter_cyc);
    }
}

int main() {
    float *out; CHECK(hipMalloc(&out, 1 << 24));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    run<1, 0>("VALU only", out);
    run<2, 0>("LDS only (step pattern)", out);
    run<2, 1>("LDS only (no broadcast)", out);
    run<3, 0>("both (step pattern)", out);
    run<3, 1>("both (no broadcast)", out);
    return 0;
}
