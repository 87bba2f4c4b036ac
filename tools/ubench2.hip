// tools/ubench2.hip -- clean VALU issue-rate probes (register operands only, no memory in the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));

// 16 independent chains; each asm block issues 16 instructions
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(1024) void valu(float *out, int iters, float seed) {
    float a[16], b[16];
    float2v pa[16], pb[16], pc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        a[k] = seed + threadIdx.x + k; b[k] = seed * k;
        pa[k] = float2v{a[k], b[k]}; pb[k] = float2v{b[k], a[k]}; pc[k] = pa[k];
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // v_add_f32
#define X(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 1) {   // v_max_f32
#define X(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 2) {   // v_max3_f32
#define X(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "v"(b[(k + 1) & 15]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 3) {   // v_pk_add_f32
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[k]) : "v"(pb[k]));
            REP16(X) REP16(X)
#undef X
        } else if (MODE == 4) {   // pk_add + max3 : 2 cells per pair (the candidate inner loop)
#define X(k) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pc[k]) : "v"(pa[k]), "v"(pb[k]));
            REP16(X)
#undef X
#define X(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(pc[k].x), "v"(pc[k].y));
            REP16(X)
#undef X
        } else if (MODE == 5) {   // add, add, max3 : 2 cells per triple
#define X(k) asm volatile("v_add_f32 %0, %1, %2" : "=v"(pc[k].x) : "v"(pa[k].x), "v"(pb[k].x));
            REP16(X)
#undef X
#define X(k) asm volatile("v_add_f32 %0, %1, %2" : "=v"(pc[k].y) : "v"(pa[k].y), "v"(pb[k].y));
            REP16(X)
#undef X
#define X(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(pc[k].x), "v"(pc[k].y));
            REP16(X)
#undef X
        } else if (MODE == 6) {   // add + max : 1 cell per pair
#define X(k) asm volatile("v_add_f32 %0, %1, %2" : "=v"(pc[k].x) : "v"(pa[k].x), "v"(pb[k].x));
            REP16(X)
#undef X
#define X(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(pc[k].x));
            REP16(X)
#undef X
        } else if (MODE == 7) {   // v_pk_max_f32 ? (probe: does it exist)  -- replaced by pk_mul as control
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[k]) : "v"(pb[k]));
            REP16(X) REP16(X)
#undef X
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k] + pa[k].x + pa[k].y + pc[k].x + pc[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F f, int reps = 4) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f(); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

template <int MODE>
void run(const char *name, int instr_per_iter, float *out, int CUS) {
    const int iters = 20000;
    for (int threads : {256, 512, 768, 1024}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(valu<MODE>, dim3(CUS), dim3(threads), 0, 0, out, iters, 1.0f); });
        double instr = (double)CUS * threads * iters * instr_per_iter;
        printf("%-22s %d waves/SIMD: %.3f ms  %.2f T lane-instr/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", name, threads / 256, ms,
               instr / ms / 1e9, (double)ms * 1e-3 * 2.4e9 / ((double)iters * instr_per_iter * (threads / 256)));
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount;
    float *out;
    CHECK(hipMalloc(&out, 1 << 24));
    run<0>("v_add_f32", 32, out, CUS);
    run<1>("v_max_f32", 32, out, CUS);
    run<2>("v_max3_f32", 32, out, CUS);
    run<3>("v_pk_add_f32", 32, out, CUS);
    run<7>("v_pk_mul_f32", 32, out, CUS);
    run<4>("pk_add+max3 (32 cells)", 32, out, CUS);
    run<5>("add,add,max3 (32 cells)", 48, out, CUS);
    run<6>("add+max (16 cells)", 32, out, CUS);
    return 0;
}
