// tools/ubench7.hip -- cost of LDS->VGPR operand delivery under the add,add,max3 stream, no DCE.
// 8x6 register tile; per iteration 96 cells (144 VALU).  NP of the 4 P float4 and NT of the 3 T float4
// are re-read from LDS every iteration into the alternate buffer (the rest are perturbed with an empty asm).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

struct Frag { float4 p[4]; float4 t[3]; };

template <int NP, int NT>
__device__ __forceinline__ void load(Frag &f, const float *base, int it, const float *pb = nullptr, const float *tb = nullptr) {
    if (pb) {
#pragma unroll
        for (int n = 0; n < 4; ++n) f.p[n] = *reinterpret_cast<const float4 *>(pb + ((it + n) & 7) * 64 + (n & 1) * 32);
#pragma unroll
        for (int n = 0; n < 3; ++n) f.t[n] = *reinterpret_cast<const float4 *>(tb + ((it + n) & 7) * 48);
        return;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        if (n < NP) f.p[n] = *reinterpret_cast<const float4 *>(base + ((it + n) & 7) * 256);
        else asm volatile("" : "+v"(f.p[n].x), "+v"(f.p[n].y), "+v"(f.p[n].z), "+v"(f.p[n].w));
    }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        if (n < NT) f.t[n] = *reinterpret_cast<const float4 *>(base + 2048 + ((it + n) & 7) * 256);
        else asm volatile("" : "+v"(f.t[n].x), "+v"(f.t[n].y), "+v"(f.t[n].z), "+v"(f.t[n].w));
    }
}

__device__ __forceinline__ void cells(float (&acc)[8][6], const Frag &f) {
    const float p0[8] = {f.p[0].x, f.p[0].y, f.p[0].z, f.p[0].w, f.p[1].x, f.p[1].y, f.p[1].z, f.p[1].w};
    const float p1[8] = {f.p[2].x, f.p[2].y, f.p[2].z, f.p[2].w, f.p[3].x, f.p[3].y, f.p[3].z, f.p[3].w};
    const float t0[6] = {f.t[0].x, f.t[0].y, f.t[0].z, f.t[0].w, f.t[1].x, f.t[1].y};
    const float t1[6] = {f.t[1].z, f.t[1].w, f.t[2].x, f.t[2].y, f.t[2].z, f.t[2].w};
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a][b] = max3(acc[a][b], p0[a] + t0[b], p1[a] + t1[b]);
}

template <int NP, int NT, int PATTERN = 0>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += blockDim.x) sm[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    const float *base = sm + (wave & 3) * 4096 + lane * 4;
    const float *pb = PATTERN ? sm + (wave & 7) * 1344 + 4 * (lane & 7) : nullptr;
    const float *tb = PATTERN ? sm + (wave & 7) * 1344 + 768 + 4 * (lane >> 3) : nullptr;
    float acc[8][6];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a][b] = -1e30f;
    Frag fa, fb;
    load<4, 3>(fa, base, 0, pb, tb);
    load<4, 3>(fb, base, 1, pb, tb);
    for (int it = 0; it < iters; it += 2) {
        load<NP, NT>(fb, base, it, pb, tb);
        __builtin_amdgcn_sched_barrier(0);
        cells(acc, fa);
        __builtin_amdgcn_sched_barrier(0);
        load<NP, NT>(fa, base, it + 1, pb, tb);
        __builtin_amdgcn_sched_barrier(0);
        cells(acc, fb);
        __builtin_amdgcn_sched_barrier(0);
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

template <int NP, int NT, int PATTERN = 0>
void run(float *out) {
    const int iters = 4000;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<NP, NT, PATTERN>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int threads : {512, 768, 1024}) {
        float ms = time_ms([&] { hipLaunchKernelGGL((k<NP, NT, PATTERN>), dim3(256), dim3(threads), 65536, 0, out, iters); });
        const double cells = 256.0 * threads * 96 * iters;
        printf("pattern %d LDS float4/iter: P %d + T %d (%2d dwords) %2d waves/CU: %.3f ms  %.1f Tcell/s\n", PATTERN, NP, NT, 4 * (NP + NT), threads / 64, ms, cells / ms / 1e9);
    }
}

int main() {
    float *out; CHECK(hipMalloc(&out, 1 << 24));
    run<4, 3, 0>(out);
    run<4, 3, 1>(out);
    return 0;
}
