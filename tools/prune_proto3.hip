// tools/prune_proto3.hip -- third prototype (list entries staged per wave by LDS-DMA) of the EXACT pruned (max,+) step (see prune_proto.hip):
// lane = one next-state j x 4 items.  Every lane walks its own sorted row; each list entry costs one
// ds_read_b128 of the [prev-state][16 items] posterior tile, 4 v_add_f32 and (entries taken in pairs) 2 v_max3_f32
// per 4 candidates -- no DPP, no cross-lane traffic.  A wave = 16 next-states x 4 item groups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
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

// LDS-DMA of 16 bytes per lane: lane l lands at lds_byte_addr + 16*l.  M0 is saved/restored inside.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
constexpr int SLOT = 2048;     // one 16-entry block of 16 rows: [half][row][64 B]

template <int NWAVES>
__global__ __launch_bounds__(64 * NWAVES) void prune_step3(const float *__restrict__ P, const float *__restrict__ TT,
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
    const char *lbase = reinterpret_cast<const char *>(lds);
    const unsigned ebase = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds + (unsigned)(NB * S * 4 + wave * 2 * SLOT));
    const int eoff = NB * S * 4 + wave * 2 * SLOT + 64 * jl;       // byte offset of this lane's row in ring slot 0
    unsigned long long nblk = 0;
    for (int jb = 16 * wave; jb < JTv; jb += 16 * NWAVES) {
        const int jj = jb + jl;
        const bool jv = jj < JTv;
        const int jr = jv ? j0 + jj : j0;
        const float2 *row = sorted + (size_t)jr * SpP + 2 * g;
        // ring of four 8-entry blocks (1 KB each: [row][64 B]), three in flight ahead of the one being consumed
        auto dma = [&](int slot, int k) { glds16(row + k, ebase + slot * 1024); };
        dma(0, 0); dma(1, 8); dma(2, 16);
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
        auto test = [&](int slot) {
            const float tn = *reinterpret_cast<const float *>(lbase + eoff + slot * 1024);
            const bool more = jv && ((tn + thr[0] > best[0]) | (tn + thr[1] > best[1]) | (tn + thr[2] > best[2]) | (tn + thr[3] > best[3]));
            return __any(more);
        };
        auto consume = [&](int slot) {
            float4 e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e[u] = *reinterpret_cast<const float4 *>(lbase + eoff + slot * 1024 + 16 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 p0 = *reinterpret_cast<const float4 *>(ptile + __builtin_bit_cast(int, e[u].y));
                const float4 p1 = *reinterpret_cast<const float4 *>(ptile + __builtin_bit_cast(int, e[u].w));
                const float t0 = e[u].x, t1 = e[u].z;
                best[0] = fmaxf(fmaxf(best[0], t0 + p0.x), t1 + p1.x);
                best[1] = fmaxf(fmaxf(best[1], t0 + p0.y), t1 + p1.y);
                best[2] = fmaxf(fmaxf(best[2], t0 + p0.z), t1 + p1.z);
                best[3] = fmaxf(fmaxf(best[3], t0 + p0.w), t1 + p1.w);
            }
        };
        for (int k = 0; k < Sp; k += 32) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (mode < 2) { if (!test(0)) break; } else if (k >= 8 * BLK) break;
            dma(3, k + 24); ++nblk; consume(0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            dma(0, k + 32); consume(1);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (mode < 2) { if (!test(2)) break; }
            dma(1, k + 40); ++nblk; consume(2);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            dma(2, k + 48); consume(3);
        }
        if (jv) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (b0 + 4 * g + it < B) out[(size_t)(b0 + 4 * g + it) * S + jr] = best[it];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a DMA may still be in flight into slot 0/1
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
    const size_t lds = (size_t)NB * S * 4 + (size_t)NWAVES_ * 2 * SLOT;
    constexpr int NWV = NWAVES_;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&prune_step3<NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int mode = ABL; (void)argc; (void)argv;
    auto go = [&] { hipLaunchKernelGGL(prune_step3<NWV>, dim3(n_bt * n_jt), dim3(64 * NWV), lds, 0, dP, dT, ds, dtv, dti, dout, dblk, B, S, Sp, SpP, n_bt, JT, mode); };
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
    printf("pruned step v3 (mode %d, %d waves): %.2f us/launch, mismatches %zu, entries scanned per wave pass %.1f (tile %d x %d, LDS %zu)\n", mode, NWV,
           bestms * 1e3 / 50, bad, (double)blk * BLK / ((double)(n_bt * n_jt) * ((JT + 15) / 16) * 64), NB, JT, lds);
    return 0;
}
