// tools/prune_proto5.hip -- fifth prototype (posterior reads software-pipelined three entry pairs ahead) of the EXACT pruned (max,+) step (see prune_proto.hip):
// lane = one next-state j x 4 items.  Every lane walks its own sorted row; each list entry costs one
// ds_read_b128 of the [prev-state][16 items] posterior tile, 4 v_add_f32 and (entries taken in pairs) 2 v_max3_f32
// per 4 candidates -- no DPP, no cross-lane traffic.  A wave = 16 next-states x 4 item groups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include <type_traits>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int R = 5;
constexpr int NB = 16;
constexpr int BLK = 16;
#ifndef ABL
#define ABL 0
#endif
#ifndef NWAVES_
#define NWAVES_ 12
#endif

typedef float f4 __attribute__((ext_vector_type(4)));
struct Block { f4 e[4]; };    // 8 entries {t, byte offset}: e[u] = {t(2u), off(2u), t(2u+1), off(2u+1)}

// The block loads are issued and awaited by hand: hipcc's waitcnt pass merges the loop-carried state
// pessimistically (s_waitcnt vmcnt(0..4) at the loop head, i.e. it drains the loads issued two pairs earlier).
// Loads complete in order, so "all but the newest N" is exact; wait_block ties the registers to the wait.
__device__ __forceinline__ void load_block(Block &blk, const float2 *p) {
    f4 a, b, c, d;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
    blk.e[0] = a; blk.e[1] = b; blk.e[2] = c; blk.e[3] = d;
}
template <int N>
__device__ __forceinline__ void wait_block(Block &blk) {
    f4 a = blk.e[0], b = blk.e[1], c = blk.e[2], d = blk.e[3];
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
    blk.e[0] = a; blk.e[1] = b; blk.e[2] = c; blk.e[3] = d;
}

template <int NWAVES>
__global__ __launch_bounds__(64 * NWAVES) void prune_step5(const float *__restrict__ P, const float *__restrict__ TT,
                                                           const float2 *__restrict__ sorted, const float *__restrict__ topv,
                                                           const int *__restrict__ topi, float *__restrict__ out,
                                                           unsigned long long *__restrict__ blocks_done, int B, int S, int Sp,
                                                           int SpP, int n_bt, int JT, int mode_rt) {
    constexpr int mode = ABL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int bt = blockIdx.x % n_bt, jt = blockIdx.x / n_bt;
    const int b0 = bt * NB, j0 = jt * JT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int JTv = S - j0 < JT ? S - j0 : JT;
    {
        const int n4 = NB * (S / 4);
        for (int e0 = tid; e0 < n4; e0 += 4 * 64 * NWAVES) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 64 * NWAVES;
                const int bb = e & (NB - 1), i4 = e / NB;
                const int brow = b0 + bb < B ? b0 + bb : B - 1;
                v[u] = e < n4 ? *reinterpret_cast<const float4 *>(P + (size_t)brow * S + 4 * i4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 64 * NWAVES;
                if (e < n4) {
                    const int bb = e & (NB - 1), i4 = e / NB;
                    float *d = lds + (4 * i4) * NB + bb;
                    d[0] = v[u].x; d[NB] = v[u].y; d[2 * NB] = v[u].z; d[3 * NB] = v[u].w;
                }
            }
        }
    }
    if (mode == 1) { __syncthreads(); return; }
    __syncthreads();
    const int jl = lane >> 2, g = lane & 3;
    const char *ptile = reinterpret_cast<const char *>(lds) + 16 * g;
    unsigned long long nblk = 0;
    for (int jb = 16 * wave; jb < JTv; jb += 16 * NWAVES) {
        const int jj = jb + jl;
        const bool jv = jj < JTv;
        const int jr = jv ? j0 + jj : j0;
        const float2 *row = sorted + (size_t)jr * SpP;
        // ring of four 8-entry blocks in registers (block n consumed, n+1 complete, n+2 / n+3 in flight) and four
        // posterior register sets: the two ds_read_b128 of entry pair s+3 are issued before the cells of pair s.
        Block q[4];
        float4 pa[4], pb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) load_block(q[i], row + 8 * i);
        float best[4], thr[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int b = b0 + 4 * g + it < B ? b0 + 4 * g + it : B - 1;
            thr[it] = topv[b * (R + 1) + R];
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < R; ++r) m = fmaxf(m, topv[b * (R + 1) + r] + TT[(size_t)topi[b * (R + 1) + r] * S + jr]);
            best[it] = m;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) wait_block<0>(q[i]);
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
            pa[s2] = *reinterpret_cast<const float4 *>(ptile + __float_as_int(q[0].e[s2].y));
            pb[s2] = *reinterpret_cast<const float4 *>(ptile + __float_as_int(q[0].e[s2].w));
        }
        auto test = [&](const Block &blk) {
            const float tn = blk.e[0].x;
            const bool more = jv && ((tn + thr[0] > best[0]) | (tn + thr[1] > best[1]) | (tn + thr[2] > best[2]) | (tn + thr[3] > best[3]));
            return __any(more);
        };
        auto half = [&](auto H, int k) {          // entry pairs 8H .. 8H+7 of the 16-pair trip
            constexpr int h = decltype(H)::value;
#pragma unroll
            for (int s2 = 8 * h; s2 < 8 * h + 8; ++s2) {
                {   // issue pair s2 + 3
                    const int n = (s2 + 3) & 15;
                    if ((n & 3) == 0) wait_block<8>(q[n / 4]);        // the two younger blocks (8 loads) may stay in flight
                    pa[n & 3] = *reinterpret_cast<const float4 *>(ptile + __float_as_int(q[n / 4].e[n & 3].y));
                    pb[n & 3] = *reinterpret_cast<const float4 *>(ptile + __float_as_int(q[n / 4].e[n & 3].w));
                }
                {
                    const float t0 = q[s2 / 4].e[s2 & 3].x, t1 = q[s2 / 4].e[s2 & 3].z;
                    const float4 p0 = pa[s2 & 3], p1 = pb[s2 & 3];
                    best[0] = fmaxf(fmaxf(best[0], t0 + p0.x), t1 + p1.x);
                    best[1] = fmaxf(fmaxf(best[1], t0 + p0.y), t1 + p1.y);
                    best[2] = fmaxf(fmaxf(best[2], t0 + p0.z), t1 + p1.z);
                    best[3] = fmaxf(fmaxf(best[3], t0 + p0.w), t1 + p1.w);
                }
                // block s2/4 fully consumed (its offsets were used three pairs ago): refill it 4 blocks ahead
                if ((s2 & 3) == 3) load_block(q[s2 / 4], row + k + 8 * (s2 / 4) + 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int k = 0; k < Sp; k += 32) {
            if (mode < 2 ? !test(q[0]) : (k >= 8 * BLK)) break;
            ++nblk;
            half(std::integral_constant<int, 0>(), k);
            if (mode < 2 && !test(q[2])) break;
            ++nblk;
            half(std::integral_constant<int, 1>(), k);
        }
        // refills are still in flight into q[]: keep the registers allocated until they have landed
#pragma unroll
        for (int i = 0; i < 4; ++i) wait_block<0>(q[i]);
        if (jv) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (b0 + 4 * g + it < B) out[(size_t)(b0 + 4 * g + it) * S + jr] = best[it];
        }
    }
    if ((blockIdx.x & 63) == 0 && lane == 0) atomicAdd(blocks_done, nblk * 64);
}

int main(int argc, char **argv) {
    const int B = 512, S = 1440, Sp = (S + 15) / 16 * 16, SpP = Sp + 64;
    std::vector<float> P((size_t)B * S), T((size_t)S * S);
    srand(1);
    auto rnd = [] { return -(float)(rand() & 0xffffff) * (16.0f / 16777216.0f); };
    for (auto &x : T) x = rnd();
    for (int b = 0; b < B; ++b) for (int i = 0; i < S; ++i) P[(size_t)b * S + i] = rnd() - 0.6f * (b % 7);
    std::vector<float2> sorted((size_t)S * SpP);
    std::vector<int> idx(S);
    for (int j = 0; j < S; ++j) {
        for (int i = 0; i < S; ++i) idx[i] = i;
        const float *row = &T[(size_t)j * S];
        std::sort(idx.begin(), idx.end(), [&](int a, int b2) { return row[a] > row[b2]; });
        for (int k = 0; k < SpP; ++k) {
            float2 v;
            if (k < S) { v.x = row[idx[k]]; v.y = __builtin_bit_cast(float, idx[k] * 64); }
            else { v.x = -INFINITY; v.y = 0.f; }
            sorted[(size_t)j * SpP + k] = v;
        }
    }
    std::vector<float> topv((size_t)B * (R + 1));
    std::vector<int> topi((size_t)B * (R + 1));
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < S; ++i) idx[i] = i;
        const float *p = &P[(size_t)b * S];
        std::partial_sort(idx.begin(), idx.begin() + R + 1, idx.end(), [&](int a, int b2) { return p[a] > p[b2]; });
        for (int r = 0; r <= R; ++r) { topv[b * (R + 1) + r] = p[idx[r]]; topi[b * (R + 1) + r] = idx[r]; }
    }
    std::vector<float> TT((size_t)S * S);
    for (int j = 0; j < S; ++j) for (int i = 0; i < S; ++i) TT[(size_t)i * S + j] = T[(size_t)j * S + i];
    float *dP, *dT, *dtv, *dout; float2 *ds; int *dti; unsigned long long *dblk;
    CHECK(hipMalloc(&dP, P.size() * 4)); CHECK(hipMalloc(&dT, T.size() * 4)); CHECK(hipMalloc(&ds, sorted.size() * 8));
    CHECK(hipMalloc(&dtv, topv.size() * 4)); CHECK(hipMalloc(&dti, topi.size() * 4)); CHECK(hipMalloc(&dout, P.size() * 4));
    CHECK(hipMalloc(&dblk, 8)); CHECK(hipMemset(dblk, 0, 8));
    CHECK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dT, TT.data(), TT.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(ds, sorted.data(), sorted.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dtv, topv.data(), topv.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dti, topi.data(), topi.size() * 4, hipMemcpyHostToDevice));
    const int n_bt = (B + NB - 1) / NB, n_jt = 256 / n_bt, JT = (S + n_jt - 1) / n_jt;
    const size_t lds = (size_t)NB * S * 4;
    constexpr int NWV = NWAVES_;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&prune_step5<NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int mode = ABL; (void)argc; (void)argv;
    auto go = [&] { hipLaunchKernelGGL(prune_step5<NWV>, dim3(n_bt * n_jt), dim3(64 * NWV), lds, 0, dP, dT, ds, dtv, dti, dout, dblk, B, S, Sp, SpP, n_bt, JT, mode); };
    go(); CHECK(hipDeviceSynchronize());
    unsigned long long blk; CHECK(hipMemcpy(&blk, dblk, 8, hipMemcpyDeviceToHost));
    std::vector<float> out(P.size());
    CHECK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int b = 0; b < B; b += 37) for (int j = 0; j < S; ++j) {
        float m = -INFINITY;
        for (int i = 0; i < S; ++i) m = fmaxf(m, P[(size_t)b * S + i] + T[(size_t)j * S + i]);
        if (m != out[(size_t)b * S + j]) ++bad;
    }
    hipEvent_t a, e; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&e));
    float bestms = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(a)); for (int k = 0; k < 50; ++k) go(); CHECK(hipEventRecord(e)); CHECK(hipEventSynchronize(e));
        float ms; CHECK(hipEventElapsedTime(&ms, a, e)); if (ms < bestms) bestms = ms;
    }
    printf("pruned step v5 (mode %d, %d waves): %.2f us/launch, mismatches %zu, entries scanned per wave pass %.1f (tile %d x %d, LDS %zu)\n", mode, NWV,
           bestms * 1e3 / 50, bad, (double)blk * BLK / ((double)(n_bt * n_jt) * ((JT + 15) / 16) * 64), NB, JT, lds);
    return 0;
}
