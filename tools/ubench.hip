// tools/ubench.hip -- MI355X micro-benchmarks that size the forward-recurrence kernel design.
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip
//
// Questions answered (numbers recorded in DESIGN.md):
//  1. issue rate of the (max,+) cell: v_add_f32 + v_max_f32, v_pk_add_f32 + v_max3_f32, with the
//     posterior operand in an SGPR (pair) or a VGPR
//  2. how fast one wave / one CU / the chip can stream scalars through s_load_dwordx16 (L2-resident)
//  3. LDS broadcast-read operand rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- 1. VALU mixes --------------------------------------------------------------------------
// MODE 0: add + max (scalar posterior operand)     2 instr / cell
// MODE 1: pk_add (vgpr pair) + max3                1 instr / cell
// MODE 2: pk_add (sgpr pair operand) + max3        1 instr / cell
// MODE 3: add x2 (sgpr) + max3                     1.5 instr / cell
template <int MODE>
__global__ __launch_bounds__(256) void valu_mix(const float *__restrict__ p, float *out, int iters) {
    constexpr int R = 32;                    // transition registers (cells per inner pass)
    float tr[R];
    float acc[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
    for (int k = 0; k < R; ++k) tr[k] = out[(threadIdx.x + k * 7) & 1023];
    for (int it = 0; it < iters; ++it) {
        // scalar operands: uniform loads (compiler emits s_load), R per pass
        const float *q = p + (size_t)(it & 63) * R;
#pragma unroll
        for (int k = 0; k < R; k += 2) {
            if (MODE == 0) {
                float c0 = q[k] + tr[k], c1 = q[k + 1] + tr[k + 1];
                acc[(k >> 1) & 3] = fmaxf(acc[(k >> 1) & 3], c0);
                acc[(k >> 1) & 3] = fmaxf(acc[(k >> 1) & 3], c1);
            } else if (MODE == 1) {
                float2v a = {tr[k], tr[k + 1]};
                float2v b = {acc[0] * 0.f + q[k], acc[1] * 0.f + q[k + 1]};   // forces VGPR operand
                float2v c;
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(c) : "v"(b), "v"(a));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[(k >> 1) & 3]) : "v"(c.x), "v"(c.y));
            } else if (MODE == 2) {
                float2v a = {tr[k], tr[k + 1]};
                float2v b = {q[k], q[k + 1]};
                float2v c;
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(c) : "s"(b), "v"(a));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[(k >> 1) & 3]) : "v"(c.x), "v"(c.y));
            } else {
                float c0, c1;
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(c0) : "s"(q[k]), "v"(tr[k]));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(c1) : "s"(q[k + 1]), "v"(tr[k + 1]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[(k >> 1) & 3]) : "v"(c0), "v"(c1));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// ---- 2. scalar streaming ----------------------------------------------------------------------
// every wave streams `bytes_per_wave` of L2-resident data through s_load_dwordx16 and consumes each
// scalar in one v_add (CONSUME=1) or only touches the first of 16 (CONSUME=0, pure fetch rate)
template <int CONSUME>
__global__ __launch_bounds__(256) void sload_stream(const float *__restrict__ p, float *out,
                                                    int floats_per_wave, int passes, int region_floats) {
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int base = (int)(((long long)wave * floats_per_wave) % region_floats);
    float tr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) tr[k] = (float)(threadIdx.x + k);
    float acc0 = -1e30f, acc1 = -1e30f;
    for (int ps = 0; ps < passes; ++ps) {
        for (int i = 0; i < floats_per_wave; i += 16) {
            const float *q = p + base + i;
            if (CONSUME) {
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    float c0 = q[k] + tr[k], c1 = q[k + 1] + tr[k + 1];
                    acc0 = fmaxf(acc0, fmaxf(c0, c1));
                }
            } else {
                acc1 = fmaxf(acc1, q[0] + tr[0]);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1;
}

// ---- 3. LDS broadcast operand -------------------------------------------------------------------
template <int REUSE>
__global__ __launch_bounds__(256) void lds_bcast(const float *__restrict__ p, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = p[i];
    __syncthreads();
    float tr[REUSE][4];
    float acc[REUSE];
#pragma unroll
    for (int r = 0; r < REUSE; ++r) {
        acc[r] = -1e30f;
#pragma unroll
        for (int k = 0; k < 4; ++k) tr[r][k] = (float)(threadIdx.x * (r + 1) + k);
    }
    const int wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int i = 0; i < 512; i += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(&sm[wave * 2048 + ((i + it * 4) & 2047)]);
#pragma unroll
            for (int r = 0; r < REUSE; ++r) {
                float c0 = v.x + tr[r][0], c1 = v.y + tr[r][1], c2 = v.z + tr[r][2], c3 = v.w + tr[r][3];
                acc[r] = fmaxf(acc[r], fmaxf(c0, c1));
                acc[r] = fmaxf(acc[r], fmaxf(c2, c3));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < REUSE; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    f();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(a));
        f();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int CUS = prop.multiProcessorCount;
    float *p, *out;
    const size_t NF = 64 << 20;
    CHECK(hipMalloc(&p, NF * 4));
    CHECK(hipMalloc(&out, 1 << 24));
    CHECK(hipMemset(p, 0, NF * 4));
    CHECK(hipMemset(out, 0, 1 << 24));

    const int iters = 20000;
    for (int wpc = 1; wpc <= 2; ++wpc) {
        const int grid = CUS * wpc;
        const double cells = (double)grid * 256 * iters * 32;
        float ms;
        ms = time_ms([&] { hipLaunchKernelGGL(valu_mix<0>, dim3(grid), dim3(256), 0, 0, p, out, iters); });
        printf("valu add+max      (2.0 i/c) %d WG/CU: %.3f ms  %.2f Tcell/s  %.2f T lane-instr/s\n", wpc, ms, cells / ms / 1e9, cells * 2 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(valu_mix<1>, dim3(grid), dim3(256), 0, 0, p, out, iters); });
        printf("valu pk_add(v)+max3 (1.0 i/c) %d WG/CU: %.3f ms  %.2f Tcell/s\n", wpc, ms, cells / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(valu_mix<2>, dim3(grid), dim3(256), 0, 0, p, out, iters); });
        printf("valu pk_add(s)+max3 (1.0 i/c) %d WG/CU: %.3f ms  %.2f Tcell/s\n", wpc, ms, cells / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(valu_mix<3>, dim3(grid), dim3(256), 0, 0, p, out, iters); });
        printf("valu 2add(s)+max3 (1.5 i/c) %d WG/CU: %.3f ms  %.2f Tcell/s\n", wpc, ms, cells / ms / 1e9);
    }

    // scalar streaming: region = 368 KB (one XCD's posterior tile) .. 8 MB .. 64 MB
    for (int region_kb : {368, 8192, 65536}) {
        const int region_floats = region_kb * 256;
        for (int wpc = 1; wpc <= 2; ++wpc) {
            const int grid = CUS * wpc;
            const int fpw = 23040;   // 92 KB per wave per pass
            const int passes = 40;
            const double bytes = (double)grid * 4 * fpw * 4.0 * passes;
            float ms = time_ms([&] { hipLaunchKernelGGL(sload_stream<1>, dim3(grid), dim3(256), 0, 0, p, out, fpw, passes, region_floats); });
            printf("sload consume region %6d KB %d WG/CU: %.3f ms  %.2f TB/s chip  %.2f B/clk/CU  %.2f Tcell/s\n", region_kb, wpc, ms,
                   bytes / ms / 1e9, bytes / ms / 1e6 / CUS / 2.4e3, bytes / 4 / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(sload_stream<0>, dim3(grid), dim3(256), 0, 0, p, out, fpw, passes, region_floats); });
            printf("sload fetch   region %6d KB %d WG/CU: %.3f ms  %.2f TB/s chip  %.2f B/clk/CU\n", region_kb, wpc, ms,
                   bytes / ms / 1e9, bytes / ms / 1e6 / CUS / 2.4e3);
        }
    }

    {
        const int it2 = 2000;
        for (int wpc = 1; wpc <= 2; ++wpc) {
            const int grid = CUS * wpc;
            double cells = (double)grid * 256 * it2 * 512;
            float ms = time_ms([&] { hipLaunchKernelGGL(lds_bcast<1>, dim3(grid), dim3(256), 0, 0, p, out, it2); });
            printf("lds bcast reuse1 %d WG/CU: %.3f ms %.2f Tcell/s\n", wpc, ms, cells * 1 / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(lds_bcast<2>, dim3(grid), dim3(256), 0, 0, p, out, it2); });
            printf("lds bcast reuse2 %d WG/CU: %.3f ms %.2f Tcell/s\n", wpc, ms, cells * 2 / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(lds_bcast<4>, dim3(grid), dim3(256), 0, 0, p, out, it2); });
            printf("lds bcast reuse4 %d WG/CU: %.3f ms %.2f Tcell/s\n", wpc, ms, cells * 4 / ms / 1e9);
        }
    }
    return 0;
}
