// tools/ubench8.hip -- does v_max3_f32 overlap with v_add_f32 when they are interleaved?
// 16 independent cells per block; orderings: grouped (16 add, 16 add, 16 max3), interleaved triples,
// skewed (max3 of the previous cell after the adds of the next).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define A0(k) "v_add_f32 %[c" #k "], %[p" #k "], %[t" #k "]\n\t"
#define A1(k) "v_add_f32 %[d" #k "], %[q" #k "], %[u" #k "]\n\t"
#define M(k) "v_max3_f32 %[a" #k "], %[a" #k "], %[c" #k "], %[d" #k "]\n\t"

#define OPS(k) [a##k] "+v"(a[k]), [c##k] "=&v"(c[k]), [d##k] "=&v"(d[k])
#define INS(k) [p##k] "v"(p[k]), [t##k] "v"(t[k]), [q##k] "v"(q[k]), [u##k] "v"(u[k])

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float seed) {
    float a[8], c[8], d[8], p[8], t[8], q[8], u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = -1e30f; p[i] = seed + threadIdx.x + i; t[i] = seed * i; q[i] = p[i] * 0.5f; u[i] = t[i] + 1.0f; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile(A0(0) A0(1) A0(2) A0(3) A0(4) A0(5) A0(6) A0(7) A1(0) A1(1) A1(2) A1(3) A1(4) A1(5) A1(6) A1(7)
                         M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
                         : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7)
                         : INS(0), INS(1), INS(2), INS(3), INS(4), INS(5), INS(6), INS(7));
        } else if (MODE == 1) {
            asm volatile(A0(0) A1(0) M(0) A0(1) A1(1) M(1) A0(2) A1(2) M(2) A0(3) A1(3) M(3) A0(4) A1(4) M(4) A0(5) A1(5) M(5)
                         A0(6) A1(6) M(6) A0(7) A1(7) M(7)
                         : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7)
                         : INS(0), INS(1), INS(2), INS(3), INS(4), INS(5), INS(6), INS(7));
        } else {
            asm volatile(A0(0) A1(0) A0(1) A1(1) M(0) A0(2) A1(2) M(1) A0(3) A1(3) M(2) A0(4) A1(4) M(3) A0(5) A1(5) M(4)
                         A0(6) A1(6) M(5) A0(7) A1(7) M(6) M(7)
                         : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7)
                         : INS(0), INS(1), INS(2), INS(3), INS(4), INS(5), INS(6), INS(7));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + c[i] + d[i];
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
void run(const char *name, float *out) {
    const int iters = 40000;
    for (int threads : {512, 1024}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f); });
        printf("%-22s %d waves/SIMD: %.3f ms  %.2f T lane-instr/s  %.2f Tcell/s\n", name, threads / 256, ms,
               (double)256 * threads * iters * 24 / ms / 1e9, (double)256 * threads * iters * 16 / ms / 1e9);
    }
}

int main() {
    float *out; CHECK(hipMalloc(&out, 1 << 24));
    run<0>("grouped", out);
    run<1>("interleaved triples", out);
    run<2>("skewed triples", out);
    return 0;
}
