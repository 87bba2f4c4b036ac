// tools/ubench9.hip -- issue rates of integer min/max forms (candidates for a cheaper max on float bit
// patterns when every candidate is <= 0) and of the add,add,<3-input-min> mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float seed) {
    float a[16], b[16], c[16], d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x + i; b[i] = seed * i + 1; c[i] = a[i] - b[i]; d[i] = b[i] * 3; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 1) {
#define X(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 2) {
#define X(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 3) {
#define X(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 4) {   // add, add, min3_u32 (grouped 16)
#define X(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(c[i]) : "v"(a[i]), "v"(b[i]));
            REP16(X)
#undef X
#define X(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d[i]) : "v"(b[i]), "v"(a[i]));
            REP16(X)
#undef X
#define X(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c[i]), "v"(d[i]));
            REP16(X)
#undef X
        } else if (MODE == 5) {   // v_cmp + v_cndmask
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            REP16(X)
#undef X
        } else if (MODE == 6) {
#define X(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 7) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 8) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 9) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
            REP16(X) REP16(X)
#undef X
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + c[i] + d[i];
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
        float ms = time_ms([&] { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f); });
        printf("%-24s %d waves/SIMD: %.3f ms  %.2f T lane-instr/s\n", name, threads / 256, ms, (double)256 * threads * iters * ipi / ms / 1e9);
    }
}

int main() {
    float *out; CHECK(hipMalloc(&out, 1 << 24));
    run<0>("v_min_u32", 32, out);
    run<1>("v_min3_u32", 32, out);
    run<2>("v_max_i32", 32, out);
    run<3>("v_max3_i32", 32, out);
    run<7>("v_and_b32", 32, out);
    run<8>("v_fma_f32", 32, out);
    run<9>("v_med3_f32", 32, out);
    run<6>("v_pk_max_i16", 32, out);
    run<5>("v_cmp_gt+v_cndmask", 32, out);
    run<4>("add,add,min3_u32", 48, out);
    return 0;
}
