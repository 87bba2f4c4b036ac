// tools/ubench5.hip -- marginal cost of LDS->VGPR traffic under a saturated add,add,max3 stream.
// Each iteration: 96 cells/lane (144 VALU) + NL ds_read instructions of width WB bytes per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <int NL, int WB>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += blockDim.x) sm[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    const float *base = sm + (wave & 7) * 1024 + lane * (WB / 4);
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
    float ld[NL > 0 ? NL * (WB / 4) : 1];
    for (int it = 0; it < iters; ++it) {
        // issue the loads
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const float *src = base + ((n + it) & 7) * 64 * (WB / 4) / (WB / 4) * (WB / 4);
            if (WB == 16) { const float4 v = *reinterpret_cast<const float4 *>(src); ld[4*n] = v.x; ld[4*n+1] = v.y; ld[4*n+2] = v.z; ld[4*n+3] = v.w; }
            else if (WB == 8) { const float2 v = *reinterpret_cast<const float2 *>(src); ld[2*n] = v.x; ld[2*n+1] = v.y; }
            else { ld[n] = *src; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) acc[a][b] = max3(acc[a][b], p0[a] + t0[b], p1[a] + t1[b]);
        __builtin_amdgcn_sched_barrier(0);
        // consume: fold loaded values into the operands so nothing is loop-invariant
        if (NL > 0) {
#pragma unroll
            for (int a = 0; a < 8; ++a) { p0[a] = ld[a % (NL * (WB / 4))]; }
            t0[0] = ld[(NL * (WB / 4)) - 1];
        } else {
#pragma unroll
            for (int a = 0; a < 8; ++a) asm volatile("" : "+v"(p0[a]));
        }
    }
    float sink = 0.f;
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

template <int NL, int WB>
void run(float *out) {
    const int iters = 4000;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<NL, WB>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int threads : {512, 1024}) {
        float ms = time_ms([&] { hipLaunchKernelGGL((k<NL, WB>), dim3(256), dim3(threads), 65536, 0, out, iters); });
        printf("NL=%2d x %2dB (%3d dwords/lane/iter) %2d waves/CU: %.3f ms  %.1f ns/iter/wave-pair\n", NL, WB, NL * WB / 4, threads / 64, ms,
               ms * 1e6 / iters / (threads / 512));
    }
}

int main() {
    float *out; CHECK(hipMalloc(&out, 1 << 24));
    run<0, 16>(out);
    run<2, 16>(out);
    run<4, 16>(out);
    run<7, 16>(out);
    run<14, 8>(out);
    run<14, 4>(out);
    run<28, 4>(out);
    return 0;
}
