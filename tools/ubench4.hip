// tools/ubench4.hip -- throughput of DPP-sourced v_add_f32 (row_ror) and the dpp-add,dpp-add,max3 mix;
// also prints the lane mapping of row_ror.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

__global__ void map_kernel(int *out) {
    int v = threadIdx.x;
    int r1, r5;
    asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r1) : "v"(v));
    asm volatile("v_mov_b32_dpp %0, %1 row_ror:5 row_mask:0xf bank_mask:0xf" : "=v"(r5) : "v"(v));
    out[threadIdx.x] = r1;
    out[64 + threadIdx.x] = r5;
}

template <int MODE>
__global__ __launch_bounds__(1024) void valu(float *out, int iters, float seed) {
    float a[16], b[16], c0[16], c1[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = seed + threadIdx.x + k; b[k] = seed * k + threadIdx.x; c0[k] = 0; c1[k] = 0; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // plain add
#define X(k) asm volatile("v_add_f32 %0, %1, %2" : "=v"(c0[k]) : "v"(a[k]), "v"(b[k]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 1) {   // dpp add
#define X(k) asm volatile("v_add_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf" : "=v"(c0[k]) : "v"(a[k]), "v"(b[k]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 2) {   // dpp add, dpp add, max3
#define X(k) asm volatile("v_add_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf" : "=v"(c0[k]) : "v"(a[k]), "v"(b[k]));
            REP16(X)
#undef X
#define X(k) asm volatile("v_add_f32_dpp %0, %1, %2 row_ror:7 row_mask:0xf bank_mask:0xf" : "=v"(c1[k]) : "v"(b[k]), "v"(a[k]));
            REP16(X)
#undef X
#define X(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c0[k]), "v"(c1[k]));
            REP16(X)
#undef X
        } else if (MODE == 3) {   // add, add, max3 (reference mix)
#define X(k) asm volatile("v_add_f32 %0, %1, %2" : "=v"(c0[k]) : "v"(a[k]), "v"(b[k]));
            REP16(X)
#undef X
#define X(k) asm volatile("v_add_f32 %0, %1, %2" : "=v"(c1[k]) : "v"(b[k]), "v"(a[k]));
            REP16(X)
#undef X
#define X(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c0[k]), "v"(c1[k]));
            REP16(X)
#undef X
        } else if (MODE == 4) {   // quad_perm broadcast add
#define X(k) asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "=v"(c0[k]) : "v"(a[k]), "v"(b[k]));
            REP16(X) REP16(X)
#undef X
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k] + c0[k] + c1[k];
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

template <int MODE>
void run(const char *name, int ipi, float *out) {
    const int iters = 20000;
    for (int threads : {512, 1024}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(valu<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f); });
        printf("%-26s %d waves/SIMD: %.3f ms  %.2f T lane-instr/s\n", name, threads / 256, ms, (double)256 * threads * iters * ipi / ms / 1e9);
    }
}

int main() {
    float *out; CHECK(hipMalloc(&out, 1 << 24));
    int *m; CHECK(hipMalloc(&m, 128 * 4));
    hipLaunchKernelGGL(map_kernel, dim3(1), dim3(64), 0, 0, m);
    int h[128]; CHECK(hipMemcpy(h, m, sizeof(h), hipMemcpyDeviceToHost));
    printf("row_ror:1 lane<-src: "); for (int i = 0; i < 20; ++i) printf("%d<-%d ", i, h[i]); printf("\n");
    printf("row_ror:5 lane<-src: "); for (int i = 0; i < 20; ++i) printf("%d<-%d ", i, h[64 + i]); printf("\n");
    run<0>("v_add_f32", 32, out);
    run<1>("v_add_f32_dpp row_ror", 32, out);
    run<4>("v_add_f32_dpp quad_perm", 32, out);
    run<3>("add,add,max3", 48, out);
    run<2>("dppadd,dppadd,max3", 48, out);
    return 0;
}
