// pruned_forward.hpp -- the EXACT forward recurrence that does not evaluate every (prev, next) cell: what every pruned
// route shares (the time-resident kernel in both forms, resident_forward.hpp; the sorted-row scan for a handful of
// sequences, small_batch_forward.hpp; the backtrace over the sorted rows, lazy_backtrace.hpp).
//
// The transition matrix is the same for every batch item and timestep, so each row is sorted ONCE per matrix
// (descending).  For item b and next state j
//     m = max_i fl(post[b,i] + trans[j,i])                                   (viterbi.cpp:81-104)
// is found by (1) seeding `best` with the largest posterior(s) of the item (explicit candidates) and (2) scanning row j
// in descending transition order.  Every candidate not yet examined has post <= thr (the next-largest posterior) and
// trans <= t_k (the current list entry), hence fl(post + trans) <= fl(thr + t_k) by monotonicity of rounding; once that
// bound is <= best the maximum is final.  Only VALUES are needed here (the backpointer is recomputed along the decoded
// path by lazy_backtrace.hpp), so ties need no care and the result is bit-identical to the dense scan.  On the
// uniform-random benchmark ~13 % of the cells are examined; on banded matrices far fewer.  Worst case (nothing
// prunable) every cell is examined at a higher cost per cell than the dense kernel -- the host layer selects the path.
//
// Lanes of the tile kernels: one next-state x 4 batch items per lane, 16 next-states x 4 item groups per wave.  Every
// next-state walks its own sorted row in 16-entry blocks (two blocks ping-pong in registers, four entries per lane of the
// quad, handed round by DPP quad_perm); a list entry costs one ds_read_b128 of the [prev-state][16 items] posterior
// tile, 4 v_add_f32 and, entries taken in pairs, 2 v_max3_f32 per 4 candidates.
//
// (Rounds 1-3 also had this recurrence as ONE LAUNCH PER TIMESTEP over (batch tile x state tile) workgroups,
// step_pruned_kernel: 19.3-20.8 us per 512 x 1440 timestep against 14.9 in the cluster form and 8.6 per batch in launch
// groups.  AUTO stopped routing anything to it in round 3 and it was removed in round 4; HISTORY.md 4.3 has its design and
// measurements, `git show 8a974c8:torbi_amd/csrc/pruned_forward.hpp` the code.)
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
constexpr int kR = TORBI_KR;  // explicit top candidates per item with whole tiles; thr = (kR+1)-th largest posterior
constexpr int kNB = 16;      // batch items per tile for S <= 2048 (4 item groups per next-state); 8 above (2 groups)
constexpr int kBlk = 16;     // list entries per termination test
constexpr int kPad = 4 * kBlk;  // (-inf) entries after every list row: prefetches never leave the row
constexpr int kTop = kR + 1;
constexpr int kMaxS16 = 2048; // the posterior tile [S][16 items] fp32 must leave room in the 160 KB LDS
constexpr int kMaxS = 4096;   // [S][8 items] tiles above kMaxS16
constexpr int kStatSlots = 64;   // scan statistics: [0,64) list blocks walked, [64,128) passes counted

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

}  // namespace pruned
