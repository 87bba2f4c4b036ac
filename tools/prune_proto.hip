// tools/prune_proto.hip -- prototype of an EXACT pruned (max,+) step:  m[b][j] = max_i (P[b][i] + T[j][i])
// using per-row descending-sorted transition lists (batch independent) and per-item top-R posteriors.
//   seed    best = max_{r<R} P_(r) + T[j][i_(r)]
//   scan    pairs (t_k, i_k) of row j in descending t; every unexamined candidate is <= t_k + P_(R+1),
//           so the scan stops (per wave) once t_k + thr <= best on every lane.
// Lanes: 16 items x 4 rows per wave; 16 consecutive pairs of a row are loaded by the 16 lanes of a row group
// with one coalesced dwordx2 each and consumed by DPP row rotation (max is order independent).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int R = 5;          // explicit top candidates per item; thr = (R+1)-th largest posterior
constexpr int NB = 16;        // items per tile
constexpr int LOOK = 4;       // 16-pair blocks in flight per row group
#ifndef NWAVES_
#define NWAVES_ 15
#endif

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }

// one rotation step: every lane consumes the pair held by lane (c + N) % 16 of its row group.
// The list stores the prev-state as a byte offset (i * 64) into the [i][16 items] posterior tile, so the
// gather address is one DPP add (offset + 4*c) and the candidate one DPP add (t + P).
template <int N>
__device__ __forceinline__ void rot_step(float &best, float pt, int poff, int cbytes, const char *tile) {
    float tt;
    int addr;
    if (N == 0) {
        tt = pt;
        addr = poff + cbytes;
    } else {
        asm("v_add_u32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "=v"(addr) : "v"(poff), "v"(cbytes), "i"(N));
        tt = dpp_f<0x120 + (N == 0 ? 1 : N)>(pt);
    }
    best = fmaxf(best, *reinterpret_cast<const float *>(tile + addr) + tt);
}
#define ROT_STEP(N) rot_step<N>(best, pt, pi, cbytes, ptile);

__device__ __forceinline__ void glds4(const float *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

struct QuadPrefetch {      // everything a quad needs from global memory, issued one quad ahead
    float seed[R];
    float2 pf[LOOK];
    float first[LOOK];     // largest t of each block (the row group's lane-0 pair), broadcast load
};

template <int NWAVES>
__global__ __launch_bounds__(64 * NWAVES) void prune_step(const float *__restrict__ P, const float *__restrict__ TT,
                                                  const float2 *__restrict__ sorted, const float *__restrict__ topv,
                                                  const int *__restrict__ topi, float *__restrict__ out,
                                                  unsigned long long *__restrict__ blocks_done, int B, int S, int Sp,
                                                  int n_bt, int JT, int mode) {
    extern __shared__ float lds[];
    const int ld = S + 1;
    const int tile = blockIdx.x;
    const int bt = tile % n_bt, jt = tile / n_bt;
    const int b0 = bt * NB, j0 = jt * JT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, js = lane >> 4;
    const int b = b0 + c;
    const int bq = b < B ? b : B - 1;
    const int nquads = mode == 1 ? 0 : (JT + 3) / 4;
    const int nb = mode == 2 ? 0 : Sp / 16;

    float tv[R];
    int ti[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { tv[r] = topv[bq * (R + 1) + r]; ti[r] = topi[bq * (R + 1) + r]; }
    const float thr = topv[bq * (R + 1) + R];

    auto issue = [&](QuadPrefetch &pre, int q) {
        const int j = j0 + 4 * q + js;
        const int jr = (4 * q + js < JT && j < S) ? j : 0;
#pragma unroll
        for (int r = 0; r < R; ++r) pre.seed[r] = TT[(size_t)ti[r] * S + jr];   // T[j][i_r] from the transposed copy
        const float2 *row = sorted + (size_t)jr * Sp + c;
#pragma unroll
        for (int u = 0; u < LOOK; ++u) { pre.pf[u] = row[16 * u]; pre.first[u] = row[16 * u - c].x; }
    };
    QuadPrefetch cur, nxt;
    if (wave < nquads) issue(cur, wave);           // in flight while the posterior rows are staged

    // stage the 16 posterior rows: float4 global loads issued in batches, scalar LDS stores (odd stride)
    {
        const int n4 = NB * (S / 4);
        for (int e0 = tid; e0 < n4; e0 += 4 * 64 * NWAVES) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 64 * NWAVES;
                const int bb = e & (NB - 1), i4 = e / NB;     // lanes: 16 rows x 4 float4 columns
                const int brow = b0 + bb < B ? b0 + bb : B - 1;
                v[u] = e < n4 ? *reinterpret_cast<const float4 *>(P + (size_t)brow * S + 4 * i4) : make_float4(0, 0, 0, 0);
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
    __syncthreads();
    const char *ptile = reinterpret_cast<const char *>(lds);
    const int cbytes = 4 * c;
    unsigned long long nblk = 0;
    for (int q = wave; q < nquads; q += NWAVES) {
        if (q + NWAVES < nquads) issue(nxt, q + NWAVES);
        const int j = j0 + 4 * q + js;
        const bool jv = 4 * q + js < JT && j < S;
        const int jr = jv ? j : 0;
        float best = -INFINITY;
#pragma unroll
        for (int r = 0; r < R; ++r) best = fmaxf(best, tv[r] + cur.seed[r]);
        const float2 *row = sorted + (size_t)jr * Sp + c;
        for (int kb = 0; kb < nb; kb += LOOK) {
#pragma unroll
            for (int u = 0; u < LOOK; u += 2) {
                // two 16-pair blocks per test: stop once no lane's bound t_first + thr exceeds its best
                if (!__any(jv && cur.first[u] + thr > best) || kb + u >= nb) goto done;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float pt = cur.pf[u + h].x;
                    const int pi = __builtin_bit_cast(int, cur.pf[u + h].y);
                    if (kb + u + h + LOOK < nb) {
                        cur.pf[u + h] = row[16 * (kb + u + h + LOOK)];
                        cur.first[u + h] = row[16 * (kb + u + h + LOOK) - c].x;
                    }
                    ++nblk;
                    ROT_STEP(0) ROT_STEP(1) ROT_STEP(2) ROT_STEP(3) ROT_STEP(4) ROT_STEP(5) ROT_STEP(6) ROT_STEP(7)
                    ROT_STEP(8) ROT_STEP(9) ROT_STEP(10) ROT_STEP(11) ROT_STEP(12) ROT_STEP(13) ROT_STEP(14) ROT_STEP(15)
                }
            }
        }
    done:
        if (jv && b < B) out[(size_t)b * S + j] = best;
        cur = nxt;
    }
    if (mode >= 0 && (blockIdx.x & 63) == 0 && lane == 0) atomicAdd(blocks_done, nblk * 64);   // sampled: a single counter serialises ~12 ns per atomic
}

int main(int argc, char **argv) {
    const int B = 512, S = 1440, Sp = (S + 15) / 16 * 16;
    std::vector<float> P((size_t)B * S), T((size_t)S * S);
    srand(1);
    auto rnd = [] { return -(float)(rand() & 0xffffff) * (16.0f / 16777216.0f); };
    for (auto &x : T) x = rnd();
    // realistic posteriors: obs + a slowly varying offset
    for (int b = 0; b < B; ++b) for (int i = 0; i < S; ++i) P[(size_t)b * S + i] = rnd() - 0.6f * (b % 7);
    std::vector<float2> sorted((size_t)S * Sp);
    std::vector<int> idx(S);
    for (int j = 0; j < S; ++j) {
        for (int i = 0; i < S; ++i) idx[i] = i;
        const float *row = &T[(size_t)j * S];
        std::sort(idx.begin(), idx.end(), [&](int a, int b2) { return row[a] > row[b2]; });
        for (int k = 0; k < Sp; ++k) {
            float2 v;
            if (k < S) { v.x = row[idx[k]]; v.y = __builtin_bit_cast(float, idx[k] * 64); }
            else { v.x = -INFINITY; v.y = 0.f; }
            sorted[(size_t)j * Sp + k] = v;
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
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&prune_step<NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    auto go = [&] { hipLaunchKernelGGL(prune_step<NWV>, dim3(n_bt * n_jt), dim3(64 * NWV), lds, 0, dP, dT, ds, dtv, dti, dout, dblk, B, S, Sp, n_bt, JT, mode); };
    go(); CHECK(hipDeviceSynchronize());
    unsigned long long blk; CHECK(hipMemcpy(&blk, dblk, 8, hipMemcpyDeviceToHost));
    std::vector<float> out(P.size());
    CHECK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    // exactness on a sample of items
    size_t bad = 0;
    for (int b = 0; b < B; b += 37) for (int j = 0; j < S; ++j) {
        float m = -INFINITY;
        for (int i = 0; i < S; ++i) m = fmaxf(m, P[(size_t)b * S + i] + T[(size_t)j * S + i]);
        if (m != out[(size_t)b * S + j]) ++bad;
    }
    hipEvent_t a, e; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&e));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(a)); for (int k = 0; k < 50; ++k) go(); CHECK(hipEventRecord(e)); CHECK(hipEventSynchronize(e));
        float ms; CHECK(hipEventElapsedTime(&ms, a, e)); if (ms < best) best = ms;
    }
    printf("pruned step: %.2f us/launch, mismatches %zu, 16-pair blocks per row-quad %.1f (tile %d x %d, LDS %zu)\n", best * 1e3 / 50, bad,
           (double)blk / ((double)(n_bt * n_jt) * ((JT + 3) / 4)), NB, JT, lds);
    return 0;
}
