// pruned_forward.hpp -- EXACT forward recurrence that does not evaluate every (prev, next) cell.
//
// The transition matrix is the same for every batch item and timestep, so each row is sorted ONCE
// per decode (descending).  For item b and next state j
//     m = max_i fl(post[b,i] + trans[j,i])                                   (viterbi.cpp:81-104)
// is found by (1) seeding `best` with the R largest posteriors of the item (explicit candidates) and
// (2) scanning row j in descending transition order.  Every candidate not yet examined has
// post <= thr (the (R+1)-th largest posterior) and trans <= t_k (the current list entry), hence
// fl(post + trans) <= fl(thr + t_k) by monotonicity of rounding; once that bound is <= best the
// maximum is final.  Only VALUES are needed here (the backpointer is recomputed along the decoded
// path by lazy_backtrace.hpp), so ties need no care and the result is bit-identical to the dense
// scan.  On the uniform-random benchmark ~9 % of the cells are examined; on peaked posteriors or
// banded matrices far fewer.  Worst case (nothing prunable) every cell is examined at a higher cost
// per cell than the dense kernel -- torbi_hip.hip / torbi_amd/viterbi.py select the path.
//
// Lanes: one next-state x 4 batch items per lane, 16 next-states x 4 item groups per wave.  Every
// next-state walks its own sorted row in 16-entry blocks (two blocks ping-pong in registers, four
// entries per lane of the quad, handed round by DPP quad_perm); a list entry costs one
// ds_read_b128 of the [prev-state][16 items] posterior tile, 4 v_add_f32 and, entries taken in
// pairs, 2 v_max3_f32 per 4 candidates.  DESIGN.md 4.3 has the measurements; tools/prune_proto*.hip
// the alternatives that lost (DPP row rotation, LDS-DMA / ds_write staged lists, entry-major lists).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>

#include "wave_reduce.hpp"

namespace pruned {

#ifndef TORBI_KR
#define TORBI_KR 3
#endif
constexpr int kR = TORBI_KR;  // explicit top candidates per item; thr = (kR+1)-th largest posterior
constexpr int kNB = 16;      // batch items per tile for S <= 2048 (4 item groups per next-state); 8 above (2 groups)
constexpr int kBlk = 16;     // list entries per termination test
constexpr int kPad = 4 * kBlk;  // (-inf) entries after every list row: prefetches never leave the row
#ifndef PRUNED_WAVES
#define PRUNED_WAVES 12
#endif
constexpr int kWaves = PRUNED_WAVES;   // waves per workgroup (12 = 3 per SIMD: 168 VGPRs each)
constexpr int kTop = kR + 1;
constexpr int kMaxS16 = 2048; // the posterior tile [S][16 items] fp32 must leave room in the 160 KB LDS
constexpr int kMaxS = 4096;   // [S][8 items] tiles above kMaxS16
constexpr int kStatSlots = 64;   // scan statistics: [0,64) blocks on the critical path, [64,128) workgroups counted
constexpr int kMaxJT = 16;   // state tiles per batch tile (kMaxJT * kTop candidates = 4 per lane of a 16-lane row)

struct Plan {
    int NI;      // batch items per tile: 16 (S <= kMaxS16) or 8
    int n_bt;    // batch tiles of NI items
    int n_jt;    // next-state tiles
    int JT;      // next-states per tile (multiple of 4)
    int Sp;      // list length rounded up to 16
    int SpP;     // list row stride in entries: Sp + kPad all-(-inf) entries so prefetch never leaves the row
    int NPOW;    // sort width (power of two >= S)
};

// B > 16: below that the generic row kernels (one workgroup per 4 next-states and item) are at least as fast
inline bool supported(int B, int S) { return B > 16 && S >= 64 && S <= kMaxS; }

// dynamic LDS of step_pruned_kernel: posterior tile [S][NI] + merged top lists + the NI items' frame counts
// + this tile's running top lists (64-bit keys) + the workgroup's deepest scan (statistics)
inline size_t lds_bytes(int S, int NI) {
    const size_t S4 = ((size_t)S + 3) / 4 * 4;      // the tile is staged four prev-states at a time
    return sizeof(float) * (NI * S4 + 2 * NI * kTop + NI + 1) + sizeof(unsigned long long) * NI * kTop;
}

inline Plan make_plan(int B, int S, int num_cus) {
    Plan p{};
    // 8-item tiles also for batches too small to give every CU a 16-item tile (n_jt is capped at kMaxJT)
    p.NI = (S <= kMaxS16 && ((B + kNB - 1) / kNB) * kMaxJT >= num_cus) ? kNB : kNB / 2;
    p.n_bt = (B + p.NI - 1) / p.NI;
    int n_jt = num_cus / p.n_bt;
    if (n_jt < 1) n_jt = 1;
    const int tile_states = (64 / (p.NI / 4)) * kWaves;             // one next-state per lane group of NI/4 lanes
    const int min_jt = (S + tile_states - 1) / tile_states;          // one pass of the workgroup covers a tile
    if (n_jt < min_jt) n_jt = min_jt;
    if (n_jt > kMaxJT) n_jt = kMaxJT;     // the per-item top lists of all state tiles are merged by one 16-lane row
    int JT = (S + n_jt - 1) / n_jt;
    const int align = p.NI == kNB ? 4 : 8;        // row groups of the arrangement pass start at tile boundaries
    JT = (JT + align - 1) / align * align;        // stays <= tile_states whenever n_jt >= min_jt (<= 11)
    p.JT = JT;
    p.n_jt = (S + JT - 1) / JT;
    p.Sp = (S + 15) / 16 * 16;
    p.SpP = p.Sp + kPad;
    p.NPOW = 64;
    while (p.NPOW < S) p.NPOW *= 2;
    return p;
}

// ---------------------------------------------------------------------------------------
// once per decode: sort every transition row in descending order (bitonic, one workgroup per row).
// Entry = {t, byte offset of prev-state i in the posterior tile = i * row_bytes (4 bytes x items per tile)}.  grid = S, block = 256,
// dynamic LDS = NPOW * 8 bytes.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sort_rows_kernel(const float *__restrict__ trans,
                                                        float2 *__restrict__ sorted, int32_t *__restrict__ row_range,
                                                        int S, int SpP, int NPOW, int row_bytes) {
    extern __shared__ float skey[];
    int *sval = reinterpret_cast<int *>(skey + NPOW);
    __shared__ int s_lo, s_hi;
    const int j = blockIdx.x;
    const float *row = trans + (size_t)j * S;
    if (threadIdx.x == 0) { s_lo = S; s_hi = -1; }
    __syncthreads();
    int lo = S, hi = -1;
    for (int k = threadIdx.x; k < NPOW; k += 256) {
        const float v = k < S ? row[k] : -INFINITY;
        skey[k] = v;
        sval[k] = k < S ? k * row_bytes : 0;
        if (v != -INFINITY) { lo = min(lo, k); hi = max(hi, k); }
    }
    if (hi >= 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();
    // prev-states this row can reach: [lo, hi]; a row of (-inf) only claims everything (never pruned anyway)
    const bool dead = s_hi < 0;
    const int row_lo = dead ? 0 : s_lo, row_hi = dead ? S - 1 : s_hi;
    if (threadIdx.x == 0) { row_range[2 * j] = row_lo; row_range[2 * j + 1] = row_hi; }
    for (int size = 2; size <= NPOW; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int k = threadIdx.x; k < NPOW; k += 256) {
                const int partner = k ^ stride;
                if (partner > k) {
                    const bool desc = (k & size) == 0;          // descending in even blocks
                    const float a = skey[k], b = skey[partner];
                    if (desc ? (a < b) : (a > b)) {
                        skey[k] = b; skey[partner] = a;
                        const int t = sval[k]; sval[k] = sval[partner]; sval[partner] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
    float2 *out = sorted + (size_t)j * SpP;
    for (int k = threadIdx.x; k < SpP; k += 256) {
        float2 v;
        v.x = k < S ? skey[k] : -INFINITY;
        // (-inf) entries and the padding name a prev-state inside the row's range: a tile stages only the
        // posterior rows its next-states can reach, and the scan may still touch such an entry (block granularity)
        v.y = __builtin_bit_cast(float, (k < S && v.x != -INFINITY) ? sval[k] : row_lo * row_bytes);
        out[k] = v;
    }
}

// ---------------------------------------------------------------------------------------
// once per decode, after the sort: reorder the entries INSIDE every 16-entry block so that the RG rows of an
// aligned row group name prev-states of different residue mod RG at the same block position wherever possible.
// The step kernel puts a row group on one ds_read_b128 lane group, all its rows read the same position at the
// same time, and a posterior row [prev-state][NI items] is 256/RG bytes of the 256-byte bank row (RG = 4 for
// 16-item tiles, 8 for 8-item tiles): equal residues are a bank conflict (2.1 LDS cycles per read for random
// lists with RG = 4, ~1.4 after this pass).  The maximum is order independent and position 0 (the block's
// largest t, used by the termination test) stays put, so results do not change.  One thread per (row group,
// block); grid covers S/RG * SpP/16 threads.
// ---------------------------------------------------------------------------------------
template <int RG>
__global__ __launch_bounds__(64) void arrange_blocks_kernel(float2 *__restrict__ sorted, int S, int SpP) {
    constexpr int SHIFT = RG == 4 ? 6 : 5;     // log2(bytes of a posterior row)
    typedef unsigned long long u64;            // RG x 8-bit counters
    const int nblk = SpP / kBlk;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (S / RG) * nblk) return;
    const int q = id / nblk, kb = id % nblk;
    u64 present[kBlk];                      // per position: counts of the residues placed so far
    {
        const float2 *r0 = sorted + (size_t)(RG * q) * SpP + kb * kBlk;
        for (int p = 0; p < kBlk; ++p) present[p] = (u64)1 << (8 * ((__float_as_int(r0[p].y) >> SHIFT) & (RG - 1)));
    }
    for (int r = 1; r < RG; ++r) {
        float2 *row = sorted + (size_t)(RG * q + r) * SpP + kb * kBlk;
        float2 ent[kBlk], out[kBlk];
        u64 remaining = 0;                  // counts of the residues still to place
        for (int e = 0; e < kBlk; ++e) {
            ent[e] = row[e];
            if (e) remaining += (u64)1 << (8 * ((__float_as_int(ent[e].y) >> SHIFT) & (RG - 1)));
        }
        out[0] = ent[0];
        present[0] += (u64)1 << (8 * ((__float_as_int(ent[0].y) >> SHIFT) & (RG - 1)));
        unsigned used = 1u;
        for (int p = 1; p < kBlk; ++p) {
            int pick = -1, key = 1 << 30;
            for (int e = 1; e < kBlk; ++e) {
                if ((used >> e) & 1u) continue;
                const int res = (__float_as_int(ent[e].y) >> SHIFT) & (RG - 1);
                // fewest equal residues already at this position; then the residue with most entries left
                const int k = (int)((present[p] >> (8 * res)) & 0xffu) * 64 - (int)((remaining >> (8 * res)) & 0xffu);
                if (k < key) { key = k; pick = e; }
            }
            const int res = (__float_as_int(ent[pick].y) >> SHIFT) & (RG - 1);
            used |= 1u << pick;
            remaining -= (u64)1 << (8 * res);
            present[p] += (u64)1 << (8 * res);
            out[p] = ent[pick];
        }
        for (int p = 1; p < kBlk; ++p) row[p] = out[p];
    }
}

// once per decode: prev-state range [lo, hi] (in units of 4 prev-states) each state tile has to stage
__global__ __launch_bounds__(64) void tile_range_kernel(const int32_t *__restrict__ row_range,
                                                        int32_t *__restrict__ tile_range, int S, int JT) {
    const int jt = blockIdx.x, j0 = jt * JT;
    const int jend = j0 + JT < S ? j0 + JT : S;
    int lo = S, hi = 0;
    for (int j = j0 + (int)threadIdx.x; j < jend; j += 64) { lo = min(lo, row_range[2 * j]); hi = max(hi, row_range[2 * j + 1]); }
    lo = wavered::wave_min_i32(lo);
    hi = -wavered::wave_min_i32(-hi);
    if (threadIdx.x == 0) { tile_range[2 * jt] = lo / 4; tile_range[2 * jt + 1] = hi / 4; }
}

// once per decode: tt[i][j] = trans[j][i] (seed candidates are read along next-states)
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ trans, float *__restrict__ tt,
                                                        int S) {
    __shared__ float tile[32][33];
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (y0 + r < S && x0 + tx < S) ? trans[(size_t)(y0 + r) * S + x0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (x0 + r < S && y0 + tx < S) tt[(size_t)(x0 + r) * S + y0 + tx] = tile[tx][r];
}

// t = 0: history row 0 = obs[b,0,:] + initial                                 (viterbi.cpp:72-76)
__global__ __launch_bounds__(256) void init_history_kernel(const float *__restrict__ obs,
                                                           const float *__restrict__ initial,
                                                           float *__restrict__ hist, int B, int T, int S) {
    const size_t n = (size_t)B * S;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / S);
        const int i = (int)(e - (size_t)b * S);
        hist[(size_t)b * T * S + i] = obs[(size_t)b * T * S + i] + initial[i];
    }
}

// once per decode: empty partial top lists (value -inf, prev-state 0) for both parities
__global__ __launch_bounds__(256) void clear_top_kernel(float *__restrict__ topv, int32_t *__restrict__ topi, size_t n,
                                                        unsigned *__restrict__ stats) {
    if (blockIdx.x == 0 && threadIdx.x < 2 * kStatSlots) stats[threadIdx.x] = 0u;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        topv[e] = -INFINITY;
        topi[e] = 0;
    }
}

// kTop largest of the NE values each lane holds (value, tag) across the wave, values descending; ties
// take the lowest tag.  Results are wave-uniform; `emit(r, value, tag)` is called once per rank.
template <int NE, typename Emit>
__device__ __forceinline__ void wave_top(float (&v)[NE], const int (&tag)[NE], Emit emit) {
    unsigned long long picked = 0;
#pragma unroll
    for (int r = 0; r < kTop; ++r) {
        float lm = -INFINITY;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (!((picked >> e) & 1ull)) lm = fmaxf(lm, v[e]);
        const float m = wavered::wave_reduce_f32(lm, wavered::MaxOp());
        int lk = 0x7fffffff, le = 0;
#pragma unroll
        for (int e = NE - 1; e >= 0; --e)
            if (!((picked >> e) & 1ull) && tag[e] != 0x7fffffff && v[e] == m) { lk = tag[e]; le = e; }
        const int k = wavered::wave_min_i32(lk);
        if (lk == k && k != 0x7fffffff) picked |= 1ull << le;
        emit(r, k == 0x7fffffff ? -INFINITY : m, k == 0x7fffffff ? 0 : k);
    }
}

// The same selection inside one 16-lane row (four independent selections per wave, DPP row reductions only):
// kTop largest of the 16 * NE values a row holds.  `emit(r, value, tag)` runs on every lane with its row's result.
template <int NE, typename Emit>
__device__ __forceinline__ void row_top(float (&v)[NE], const int (&tag)[NE], Emit emit) {
    unsigned picked = 0;
#pragma unroll
    for (int r = 0; r < kTop; ++r) {
        float lm = -INFINITY;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (!((picked >> e) & 1u)) lm = fmaxf(lm, v[e]);
        const float m = wavered::row_reduce_f32(lm, wavered::MaxOp());
        int lk = 0x7fffffff, le = 0;
#pragma unroll
        for (int e = NE - 1; e >= 0; --e)
            if (!((picked >> e) & 1u) && tag[e] != 0x7fffffff && v[e] == m) { lk = tag[e]; le = e; }
        const int k = wavered::row_min_i32(lk);
        if (lk == k && k != 0x7fffffff) picked |= 1u << le;
        emit(r, k == 0x7fffffff ? -INFINITY : m, k == 0x7fffffff ? 0 : k);
    }
}

// once per decode: the kTop largest entries of history row 0 of every item, stored as the partial list of
// state tile 0 (parity 0).  One wave per item; NQ float4 per lane (S <= 256*NQ).  grid = B, block = 64.
template <int NQ>
__global__ __launch_bounds__(64) void top_kernel(const float *__restrict__ hist, float *__restrict__ topv,
                                                 int32_t *__restrict__ topi, int B, int T, int S) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const float *row = hist + (size_t)b * T * S;
    float v[NQ * 4];
    int tag[NQ * 4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = 4 * lane + 256 * q;
        // history rows are 16-byte aligned only when S % 4 == 0: plain loads (this kernel runs once per decode)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[4 * q + u] = i + u < S ? row[i + u] : -INFINITY;
            tag[4 * q + u] = i + u < S ? i + u : 0x7fffffff;
        }
    }
    wave_top<NQ * 4>(v, tag, [&](int r, float m, int k) {
        if (lane == 0) { topv[(size_t)b * kTop + r] = m; topi[(size_t)b * kTop + r] = k; }
    });
}

#ifdef PRUNED_STAMP
// build-time instrumentation (tools/pruned_stamps.py): per-wave cycle stamps of the last launch
constexpr int kStamps = 10;
__device__ unsigned long long g_stamps[1024 * kWaves * kStamps];
#define PSTAMP(i) st[i] = __builtin_readcyclecounter()
#else
#define PSTAMP(i)
#endif

// A 16-entry list block of one row is held by the row's FOUR lanes (item groups g = 0..3), four entries each:
// the wave loads every list byte once (duplicate lanes would quadruple the texture-path bytes, the busiest unit of
// this kernel) and the entries are handed round the quad by DPP quad_perm broadcasts when they are consumed.
template <int EPL>
struct ListBlock { float4 e[EPL / 2]; };   // EPL entries of the block held by this lane: e[h] = {t, off, t, off}

template <int EPL>
__device__ __forceinline__ void load_list_block(ListBlock<EPL> &blk, const float2 *row_g, int k) {   // row_g = row + EPL*g
#pragma unroll
    for (int h = 0; h < EPL / 2; ++h) blk.e[h] = *reinterpret_cast<const float4 *>(row_g + k + 2 * h);
}

// value held by lane O of this lane's group of G lanes (G = 4: the quad; G = 2: the lane pair)
template <int G, int O>
__device__ __forceinline__ int group_bcast(int x) {
    constexpr int ctrl = G == 4 ? O * 0x55 : (O | (O << 2) | ((2 + O) << 4) | ((2 + O) << 6));
    return __builtin_amdgcn_update_dpp(0, x, ctrl, 0xf, 0xf, true);
}
template <int G, int O>
__device__ __forceinline__ float group_bcast(float x) {
    return __builtin_bit_cast(float, group_bcast<G, O>(__builtin_bit_cast(int, x)));
}

// ---------------------------------------------------------------------------------------
// one timestep.  grid = n_bt * n_jt, block = 64 * kWaves, dynamic LDS = lds_bytes(S, JT).
//
// Per-item top lists travel between timesteps as PARTIAL lists: each state tile leaves the kTop
// largest of the outputs it produced for each of its 16 items (ptop[t & 1][jt][b][r]); the next
// timestep's tiles merge the n_jt partial lists of their items (every member of the global top
// kTop is in the top kTop of its own tile).  No separate selection kernel, no extra launch.
//
// Order inside the workgroup: merge top lists -> barrier -> issue the seed gathers, observation
// loads and first list block (their latency hides behind the tile staging) -> stage the posterior
// tile -> barrier -> scan -> outputs -> barrier -> this tile's partial top lists.
// ---------------------------------------------------------------------------------------
template <int NI, bool STATS>
__global__ __launch_bounds__(64 * kWaves) void step_pruned_kernel(
    const float *__restrict__ obs, const int32_t *__restrict__ frames, const float *__restrict__ tt,
    const float2 *__restrict__ sorted, const int32_t *__restrict__ tile_range, const float *__restrict__ ptopv_in,
    const int32_t *__restrict__ ptopi_in, float *__restrict__ ptopv_out, int32_t *__restrict__ ptopi_out,
    float *__restrict__ hist, unsigned *__restrict__ stats, int B, int T, int S, int t, int SpP, int n_bt, int n_jt,
    int JT) {
    constexpr int G = NI / 4;            // lanes per next-state (item groups of 4)
    constexpr int RW = 64 / G;           // next-states per wave
    constexpr int EPL = kBlk / G;        // list entries per lane per block
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // [NI][kTop] running top list of this tile's outputs per item, as 64-bit keys (order-preserving value bits,
    // ~next-state); 8-byte aligned because NI is even
    unsigned long long *ttop = reinterpret_cast<unsigned long long *>(lds + (size_t)NI * ((S + 3) / 4 * 4));
    float *mtopv = reinterpret_cast<float *>(ttop + NI * kTop);   // [16][kTop] merged top values of t-1
    int *mtopi = reinterpret_cast<int *>(mtopv + NI * kTop);
    int *sframes = mtopi + NI * kTop;                     // [16] frames of the tile's items (0 past the batch)
    int *sdeep = sframes + NI;                            // deepest scan (16-entry blocks) among this tile's waves
    // grid = (n_bt, n_jt): linear workgroup id = bt + n_bt * jt (the 8 state tiles of a batch tile share an XCD
    // whenever n_bt % 8 == 0)
    const int bt = blockIdx.x, jt = blockIdx.y;
    const int b0 = bt * NI, j0 = jt * JT;
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef PRUNED_STAMP
    unsigned long long st[kStamps] = {};
#endif
    PSTAMP(0);
    // the staged prev-state range is needed for the tile addresses: request it before anything else
    const int lo4 = tile_range[2 * jt], hi4 = tile_range[2 * jt + 1];
    __builtin_amdgcn_sched_barrier(0);
    int fr = 0;
    if (tid < NI) fr = b0 + tid < B ? frames[b0 + tid] : 0;     // stored to LDS after the loads are out
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int JTv = S - j0 < JT ? S - j0 : JT;                   // next-states of this tile
    const int Sp = (S + 15) / 16 * 16;

    // waves 0..3 merge top lists later; their candidates (n_jt * kTop <= 16 * kMergeNE per item, tag = prev-state:
    // equal values from different tiles cannot share a prev-state) are the first loads of the kernel
    constexpr int kMergeNE = (kMaxJT * kTop + 15) / 16;
    const int mitem = 4 * wave + (lane >> 4), ml16 = lane & 15;
    float mv[kMergeNE];
    int mtag[kMergeNE];
    if (wave < NI / 4) {
        const int bw = b0 + mitem < B ? b0 + mitem : B - 1;
#pragma unroll
        for (int e = 0; e < kMergeNE; ++e) {
            const int cand = ml16 + 16 * e;
            const bool ok = cand < n_jt * kTop;
            const unsigned src = ((unsigned)(ok ? cand / kTop : 0) * B + bw) * kTop + (ok ? cand % kTop : 0);
            mv[e] = ok ? ptopv_in[src] : -INFINITY;
            mtag[e] = ok ? ptopi_in[src] : 0x7fffffff;
        }
    }

    // lane = next-state jl of the wave's 16 x item group g (items 4g .. 4g+3 of the tile); the first list
    // blocks and the observations do not depend on anything staged below: issue them first
    // quads of lanes -> next-states so that every ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31},
    // +32) holds an aligned row quad (arrange_blocks_kernel keeps those conflict-poor)
    // (8-item tiles: lane pairs -> next-states, aligned groups of eight rows per lane group)
    const int g = lane & (G - 1);
    const int jl = G == 4 ? (int)((0xFBAE9DC873261540ull >> (4 * (lane >> 2))) & 15)
                          : (int)((0xFE7654DC32BA9810ull >> (4 * ((lane >> 1) & 15))) & 15) + (lane & 32) / 2;
    const int jj = RW * wave + jl;
    const bool jv = jj < JTv;
    const int jr = jv ? j0 + jj : j0;
    const float2 *row = sorted + (size_t)jr * SpP + EPL * g;    // this lane's share of every 16-entry block
    ListBlock<EPL> cur, nxt;
    load_list_block(cur, row, 0);
    load_list_block(nxt, row, kBlk);
    float ob[4];
    {
        // one 64-bit address, then a uniform stride per item (items past the batch re-read the last one)
        const int bfirst = b0 + 4 * g < B ? b0 + 4 * g : B - 1;
        const float *osrc = obs + ((size_t)bfirst * T + t) * S + jr;
        const size_t ostride = (size_t)T * S;
#pragma unroll
        for (int it = 0; it < 4; ++it) ob[it] = osrc[(bfirst + it < B ? it : B - 1 - bfirst) * ostride];
    }

    // every thread fetches its share of the 16 posterior rows in ONE round trip (all loads in flight before the
    // first LDS write); waves 0..3 merge the top lists while theirs are on the way
    // Only the prev-states this tile's next-states can reach are staged (tile_range: the whole range for a
    // dense matrix, the band for a banded one).
    constexpr int NCH = (NI * ((NI == 16 ? kMaxS16 : kMaxS) / 4) + 64 * kWaves - 1) / (64 * kWaves);
    const int n4 = NI * (hi4 - lo4 + 1);
    float4 pv[NCH];
    {
        // 64 * kWaves is a multiple of 16: a thread keeps its item and walks prev-states in steps of 4 * 48
        static_assert((64 * kWaves) % NI == 0, "tile staging assumes a fixed item per thread");
        const int bb = tid & (NI - 1);
        const int brow = b0 + bb < B ? b0 + bb : B - 1;
        const int ifirst = 4 * (lo4 + tid / NI);
        const float *psrc = hist + ((size_t)brow * T + (t - 1)) * S + ifirst;
        if ((S & 3) == 0) {
#pragma unroll
            for (int u = 0; u < NCH; ++u)
                if (tid + u * 64 * kWaves < n4) pv[u] = *reinterpret_cast<const float4 *>(psrc + u * (4 * 64 * kWaves / NI));
        } else {
            // S % 4 != 0: history rows are not 16-byte aligned; four plain loads, prev-states past S read as 0
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                if (tid + u * 64 * kWaves < n4) {
                    const int i = ifirst + u * (4 * 64 * kWaves / NI);
                    const float *q = psrc + u * (4 * 64 * kWaves / NI);
                    pv[u] = make_float4(i < S ? q[0] : 0.f, i + 1 < S ? q[1] : 0.f, i + 2 < S ? q[2] : 0.f, i + 3 < S ? q[3] : 0.f);
                }
            }
        }
    }
    PSTAMP(1);
    if (tid < NI) sframes[tid] = fr;
    if (tid < NI * kTop) ttop[tid] = 0ull;                // 0 = empty (every real key is > 0)
    if (STATS && tid == 0) *sdeep = 0;
    if (wave < NI / 4) {
        // every 16-lane row merges the partial top lists of one item (candidates fetched at kernel entry)
        auto emit = [&](int r, float m, int k) {
            if (ml16 == 0) { mtopv[mitem * kTop + r] = m; mtopi[mitem * kTop + r] = k; }
        };
        if (n_jt * kTop <= 32) {
            float v2[2] = {mv[0], mv[1]};
            const int t2[2] = {mtag[0], mtag[1]};
            row_top<2>(v2, t2, emit);
        } else {
            row_top<kMergeNE>(mv, mtag, emit);
        }
    }
    // tile layout [prev-state][16 items]: lanes = 16 rows x 4 float4 columns
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int e = tid + u * 64 * kWaves;
        if (e < n4) {
            const int bb = e & (NI - 1), i4 = lo4 + e / NI;
            float *d = lds + (4 * i4) * NI + bb;
            d[0] = pv[u].x; d[NI] = pv[u].y; d[2 * NI] = pv[u].z; d[3 * NI] = pv[u].w;
        }
    }
    PSTAMP(2);
    // one barrier publishes the tile, the merged lists and the frame counts; tiles whose items have all
    // ended (t >= batch_frames[b]) stop here
    if (!__syncthreads_or(t < fr)) return;
    PSTAMP(3);

    // seeds: the kR largest posteriors of each item are explicit candidates; their gathers fly while the
    // first list block is consumed (examining more candidates never changes the maximum)
    float seedv[4][kR], seedt[4][kR], thr[4];
    bool live[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int item = 4 * g + it;
        live[it] = t < sframes[item];
        thr[it] = mtopv[item * kTop + kR];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            seedv[it][r] = mtopv[item * kTop + r];
            seedt[it][r] = tt[(unsigned)(mtopi[item * kTop + r] * S + jr)];        // trans[jr][i_r]
        }
    }
    PSTAMP(4);
    const char *ptile = reinterpret_cast<const char *>(lds) + 16 * g;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    auto pair = [&](float t0, int o0, float t1, int o1) {
        // keep the broadcast t values in registers: folded into the adds they become four half-rate
        // v_add_f32_dpp per entry instead of one v_mov_b32_dpp + four full-rate v_add_f32
        asm volatile("" : "+v"(t0), "+v"(t1));
        const float4 p0 = *reinterpret_cast<const float4 *>(ptile + o0);
        const float4 p1 = *reinterpret_cast<const float4 *>(ptile + o1);
        best[0] = fmaxf(fmaxf(best[0], t0 + p0.x), t1 + p1.x);
        best[1] = fmaxf(fmaxf(best[1], t0 + p0.y), t1 + p1.y);
        best[2] = fmaxf(fmaxf(best[2], t0 + p0.z), t1 + p1.z);
        best[3] = fmaxf(fmaxf(best[3], t0 + p0.w), t1 + p1.w);
    };
    auto owner = [&](auto Oc, const ListBlock<EPL> &blk) {      // the entries held by lane O of the group
        constexpr int O = decltype(Oc)::value;
#pragma unroll
        for (int h = 0; h < EPL / 2; ++h)
            pair(group_bcast<G, O>(blk.e[h].x), group_bcast<G, O>(__float_as_int(blk.e[h].y)),
                 group_bcast<G, O>(blk.e[h].z), group_bcast<G, O>(__float_as_int(blk.e[h].w)));
    };
    auto consume = [&](const ListBlock<EPL> &blk) {
        owner(std::integral_constant<int, 0>(), blk);
        owner(std::integral_constant<int, 1>(), blk);
        if (G == 4) {
            owner(std::integral_constant<int, 2 % G>(), blk);
            owner(std::integral_constant<int, 3 % G>(), blk);
        }
    };
    int nblk = 1;                          // wave-uniform: blocks this wave examines
    consume(cur);
    load_list_block(cur, row, 2 * kBlk);
    PSTAMP(5);
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int r = 0; r < kR; ++r) best[it] = fmaxf(best[it], seedv[it][r] + seedt[it][r]);
    PSTAMP(6);
    // stop once no lane's bound fl(t_first + thr) exceeds its best (t_first = largest unexamined entry)
    auto more = [&](const ListBlock<EPL> &blk) {
        const float tn = group_bcast<G, 0>(blk.e[0].x);
        return __any(jv && ((tn + thr[0] > best[0]) | (tn + thr[1] > best[1]) | (tn + thr[2] > best[2]) |
                            (tn + thr[3] > best[3])));
    };
    // two blocks ping-pong by name (no register copies); rows carry kPad (-inf) entries past Sp.
    // Block 0 is done; `nxt` holds block 1, `cur` is being refilled with block 2.
    for (int k = kBlk; k < Sp; k += 2 * kBlk) {
        if (!more(nxt)) break;
        if (STATS) ++nblk;
        consume(nxt);
        load_list_block(nxt, row, k + 2 * kBlk);
        if (!more(cur)) break;
        if (STATS) ++nblk;
        consume(cur);
        load_list_block(cur, row, k + 3 * kBlk);
    }
    PSTAMP(7);
    // outputs, and this tile's top lists: an output enters the item's list only if it beats the list's current
    // last entry (rare once a few outputs have arrived); insertion is a cascade of 64-bit LDS atomic maxima, the
    // displaced key moving one rank down -- every rank ends with the maximum of what passed through it.
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int item = 4 * g + it;
        const float o = ob[it] + best[it];                                   // post'[j] = obs[t,j] + max
        if (jv && live[it]) hist[((size_t)(b0 + item) * T + t) * S + jr] = o;
        unsigned u = __float_as_uint(o);
        u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;                           // unsigned order == float order
        unsigned long long x = ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - jr);
        if (!jv || x <= ttop[item * kTop + kTop - 1]) x = 0ull;
#pragma unroll
        for (int r = 0; r < kTop; ++r) {
            if (x != 0ull) {
                const unsigned long long old = atomicMax(&ttop[item * kTop + r], x);
                x = old < x ? old : x;
            }
        }
    }
    const bool sampled = STATS && (t & 7) == 1;       // the STATS instance only, from every 8th timestep
    if (sampled && lane == 0) atomicMax(sdeep, nblk);
    __syncthreads();
    PSTAMP(8);
    // statistics for adaptive path selection (torbi_hip_scan_stats): the launch lasts as long as its deepest wave
    if (sampled && tid == 0) {
        const int slot = (blockIdx.x + gridDim.x * blockIdx.y) & (kStatSlots - 1);
        atomicAdd(&stats[slot], (unsigned)*sdeep);
        atomicAdd(&stats[kStatSlots + slot], 1u);
    }
    if (tid < NI * kTop) {
        const int item = tid / kTop, r = tid % kTop;
        const unsigned long long k = ttop[tid];
        unsigned u = (unsigned)(k >> 32);
        u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
        if (b0 + item < B) {
            ptopv_out[((size_t)jt * B + b0 + item) * kTop + r] = k ? __uint_as_float(u) : -INFINITY;
            ptopi_out[((size_t)jt * B + b0 + item) * kTop + r] = k ? 0x7fffffff - (int)(unsigned)k : 0;
        }
    }
#ifdef PRUNED_STAMP
    PSTAMP(9);
    const int linear = blockIdx.x + gridDim.x * blockIdx.y;
    if (lane == 0 && linear < 1024)
        for (int i = 0; i < kStamps; ++i) g_stamps[((size_t)linear * kWaves + wave) * kStamps + i] = st[i];
#endif
}

}  // namespace pruned
