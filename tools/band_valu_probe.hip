// Issue rate of the band kernel's cell arithmetic on registers alone: per "dquad" 64 v_add_f32 + 32 v_max3_f32, 12 waves per
// workgroup (3 per SIMD), one workgroup per compute unit.  Variants: the compiler's schedule / adds only / max3 only.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(768) void probe(float *out, const float *in, int iters) {
    const int tid = threadIdx.x;
    float4 w[8], t[4];
    for (int m = 0; m < 8; ++m) w[m] = reinterpret_cast<const float4 *>(in)[tid + 768 * m];
    for (int d = 0; d < 4; ++d) t[d] = reinterpret_cast<const float4 *>(in)[tid + 768 * (8 + d)];
    float acc[16];
    for (int e = 0; e < 16; ++e) acc[e] = -1e30f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 r0 = w[(2 * h + k) & 7], r1 = w[(2 * h + 1 + k) & 7];
                const float t0 = ((const float *)&t[2 * h])[k], t1 = ((const float *)&t[2 * h + 1])[k];
                if (MODE == 0) {
                    acc[4 * k + 0] = fmaxf(fmaxf(acc[4 * k + 0], r0.x + t0), r1.x + t1);
                    acc[4 * k + 1] = fmaxf(fmaxf(acc[4 * k + 1], r0.y + t0), r1.y + t1);
                    acc[4 * k + 2] = fmaxf(fmaxf(acc[4 * k + 2], r0.z + t0), r1.z + t1);
                    acc[4 * k + 3] = fmaxf(fmaxf(acc[4 * k + 3], r0.w + t0), r1.w + t1);
                } else if (MODE == 1) {      // 8 adds, no maxima (each add feeds the next one's accumulator)
                    acc[4 * k + 0] = (acc[4 * k + 0] + r0.x) + r1.x;
                    acc[4 * k + 1] = (acc[4 * k + 1] + r0.y) + r1.y;
                    acc[4 * k + 2] = (acc[4 * k + 2] + r0.z) + r1.z;
                    acc[4 * k + 3] = (acc[4 * k + 3] + r0.w) + r1.w;
                } else if (MODE == 2) {      // 4 max3, no adds
                    acc[4 * k + 0] = fmaxf(fmaxf(acc[4 * k + 0], r0.x), r1.x);
                    acc[4 * k + 1] = fmaxf(fmaxf(acc[4 * k + 1], r0.y), r1.y);
                    acc[4 * k + 2] = fmaxf(fmaxf(acc[4 * k + 2], r0.z), r1.z);
                    acc[4 * k + 3] = fmaxf(fmaxf(acc[4 * k + 3], r0.w), r1.w);
                } else {                     // 8 plain two-operand maxima
                    acc[4 * k + 0] = fmaxf(acc[4 * k + 0], r0.x); acc[4 * k + 0] = fmaxf(acc[4 * k + 0], r1.x);
                    acc[4 * k + 1] = fmaxf(acc[4 * k + 1], r0.y); acc[4 * k + 1] = fmaxf(acc[4 * k + 1], r1.y);
                    acc[4 * k + 2] = fmaxf(acc[4 * k + 2], r0.z); acc[4 * k + 2] = fmaxf(acc[4 * k + 2], r1.z);
                    acc[4 * k + 3] = fmaxf(acc[4 * k + 3], r0.w); acc[4 * k + 3] = fmaxf(acc[4 * k + 3], r1.w);
                }
            }
        }
        // rotate the window so that nothing is loop invariant
        asm volatile("" : "+v"(w[0].x), "+v"(w[1].y), "+v"(w[2].z), "+v"(w[3].w), "+v"(t[0].x), "+v"(t[1].y), "+v"(t[2].z), "+v"(t[3].w));
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc[e];
    out[blockIdx.x * 768 + tid] = s;
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 768 * 4);
    hipMalloc(&in, 768 * 12 * 16);
    hipMemset(in, 0, 768 * 12 * 16);
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const char *names[] = {"64 add + 32 max3 (the dquad)", "64 add", "32 max3", "64 max (two operands)"};
    auto run = [&](auto kernel, int mode) {
        kernel<<<256, 768>>>(out, in, 100);
        hipEventRecord(a);
        kernel<<<256, 768>>>(out, in, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-32s %.3f ms  %.1f ns per dquad of one wave (3 waves per SIMD) = %.0f cycles at 2.1 GHz\n", names[mode], ms,
               ms * 1e6 / iters / 3, ms * 1e6 / iters / 3 * 2.1);
    };
    run(probe<0>, 0);
    run(probe<1>, 1);
    run(probe<2>, 2);
    run(probe<3>, 3);
    return 0;
}
