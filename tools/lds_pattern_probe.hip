// Cycles per ds_read_b128 wave-instruction for the band kernel's access patterns (12 waves per workgroup, one workgroup
// per compute unit).  hipcc --offload-arch=gfx950 -O3 -o tools/lds_pattern_probe tools/lds_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(768) void probe(float *out, int iters, int ig_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 160 * 1024 / 4; e += 768) reinterpret_cast<float *>(lds)[e] = (float)e;
    __syncthreads();
    const int ig = lane & 3, jgl = lane >> 2, blk = wave >> 2;
    const int jg = min(16 * blk + jgl, 44);
    unsigned base;
    if (MODE == 0) base = lane * 16 + wave * 1024;                                  // 64 distinct contiguous 16-byte slots
    else if (MODE == 1) base = 126720 + (ig * ig_stride + 16 * jg) * 4;             // W: [ig][row][4 items]
    else if (MODE == 2) base = jg * 64;                                             // Tq: the 4 item groups read one address
    else if (MODE == 3) base = (lane >> 2) * 16;                                    // 16 distinct addresses, each read by a quad
    else if (MODE == 4) base = (lane & 15) * 16;                                    // 16 distinct addresses, by lane mod 16
    else base = jg * 64 + ig * 16;                                                  // Tq rows spread over the quad (distinct)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        const char *p = lds + base + (it & 7) * (MODE == 1 ? 64 : MODE == 2 || MODE == 5 ? 2880 : 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (MODE == 1 ? 16 * u : MODE == 2 || MODE == 5 ? 16 * (u & 3) + 2880 * (u >> 2) : 4096 * u));
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    out[blockIdx.x * 768 + tid] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 768 * 4);
    const int iters = 20000;
    const char *names[] = {"contiguous 16 B per lane", "W pattern (ig stride 1476)", "Tq pattern (quad reads one address)",
                           "16 addresses, one per quad", "16 addresses, lane mod 16", "Tq rows spread over the quad"};
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kernel, int mode, int stride) {
        hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        kernel<<<256, 768, 160 * 1024>>>(out, 100, stride);
        hipEventRecord(a);
        kernel<<<256, 768, 160 * 1024>>>(out, iters, stride);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double instr = 12.0 * iters * 8;       // ds_read_b128 wave-instructions per compute unit
        printf("%-40s %.3f ms  %.2f ns per wave-instruction per CU = %.2f cycles at 2.1 GHz\n", names[mode], ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.1);
    };
    run(probe<0>, 0, 0);
    run(probe<1>, 1, 1476);
    run(probe<1>, 1, 1424);
    run(probe<2>, 2, 0);
    run(probe<3>, 3, 0);
    run(probe<4>, 4, 0);
    run(probe<5>, 5, 0);
    return 0;
}
