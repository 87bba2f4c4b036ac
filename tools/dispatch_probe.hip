// tools/dispatch_probe.hip -- when a workgroup that fills a CU (93 KB LDS, 12 waves x 168 VGPRs) exits, how soon does the
// next workgroup of the SAME launch (or of a launch on another stream) get that CU?
//   hipcc --offload-arch=gfx950 -O3 -o tools/dispatch_probe tools/dispatch_probe.hip && tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// every workgroup spins for dur[blockIdx.x] microseconds (wall clock: 100 MHz) and stamps its start / end
template <int VG>
__global__ __launch_bounds__(768) void spin_kernel(const int *__restrict__ dur, unsigned long long *__restrict__ stamps, int base) {
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    float keep[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) keep[i] = lds[(threadIdx.x + i) & 1023];
    const unsigned long long ticks = (unsigned long long)dur[blockIdx.x] * 100ull;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < VG; ++i) keep[i] = keep[i] * 1.0001f + 0.5f;
        __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += keep[i];
    if (s == 12345.678f) lds[threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * (base + blockIdx.x)] = t0; stamps[2 * (base + blockIdx.x) + 1] = wall_clock64(); }
}

int main() {
    const int lds = 93 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spin_kernel<120>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int *dur; unsigned long long *stamps;
    CHECK(hipMalloc(&dur, 4096 * sizeof(int)));
    CHECK(hipMalloc(&stamps, 2 * 4096 * sizeof(unsigned long long)));
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto report = [&](const char *what, int n) {
        std::vector<unsigned long long> h(2 * n);
        hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0, second = ~0ull;
        for (int i = 0; i < n; ++i) { t0 = h[2 * i] < t0 ? h[2 * i] : t0; t1 = h[2 * i + 1] > t1 ? h[2 * i + 1] : t1; }
        for (int i = 256; i < n; ++i) second = h[2 * i] < second ? h[2 * i] : second;
        printf("%-70s total %7.2f ms; first workgroup beyond the 256th starts at %7.2f ms\n", what, (t1 - t0) / 1e5,
               n > 256 ? (second - t0) / 1e5 : 0.0);
    };
    std::vector<int> d(4096);
    // (1) one launch of 512: the first 256 take 1 ms except workgroup 0 (10 ms); the next 256 take 5 ms
    for (int i = 0; i < 512; ++i) d[i] = i == 0 ? 10000 : i < 256 ? 1000 : 5000;
    CHECK(hipMemcpy(dur, d.data(), 512 * sizeof(int), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(spin_kernel<120>, dim3(512), dim3(768), lds, s1, dur, stamps, 0);
        CHECK(hipDeviceSynchronize());
    }
    report("one launch of 512: wg 0 10 ms, wgs 1-255 1 ms, wgs 256-511 5 ms", 512);
    // (2) the same, the long workgroup LAST of the first 256
    for (int i = 0; i < 512; ++i) d[i] = i == 255 ? 10000 : i < 256 ? 1000 : 5000;
    CHECK(hipMemcpy(dur, d.data(), 512 * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(spin_kernel<120>, dim3(512), dim3(768), lds, s1, dur, stamps, 0);
    CHECK(hipDeviceSynchronize());
    report("one launch of 512: wg 255 10 ms, other first-256 1 ms, wgs 256-511 5 ms", 512);
    // (3) two launches of 256 on two streams with the durations of (1)
    for (int i = 0; i < 512; ++i) d[i] = i == 0 ? 10000 : i < 256 ? 1000 : 5000;
    CHECK(hipMemcpy(dur, d.data(), 512 * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(spin_kernel<120>, dim3(256), dim3(768), lds, s1, dur, stamps, 0);
    hipLaunchKernelGGL(spin_kernel<120>, dim3(256), dim3(768), lds, s2, dur + 256, stamps, 256);
    CHECK(hipDeviceSynchronize());
    report("two launches of 256 on two streams, same durations", 512);
    // (4) descending ramp 9..1 ms over the first 256, then 256 x 5 ms in the same launch
    for (int i = 0; i < 512; ++i) d[i] = i < 256 ? 9000 - 31 * i : 5000;
    CHECK(hipMemcpy(dur, d.data(), 512 * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(spin_kernel<120>, dim3(512), dim3(768), lds, s1, dur, stamps, 0);
    CHECK(hipDeviceSynchronize());
    report("one launch of 512: first 256 ramp 9 -> 1 ms, next 256 5 ms", 512);
    // (5) small-footprint workgroups (no LDS limit: 64 threads) for comparison are not needed: the question is the big ones
    return 0;
}
