// wide_forward.hpp -- the time-resident forward kernel (resident_forward.hpp, whole tiles) with ONE NEXT-STATE x ALL 16
// ITEMS per lane instead of one next-state x 4 items.
//
// Why.  resident_forward_kernel spends 2.4 vector instructions per examined cell: a list entry {t, prev-state} is held
// by one lane of the quad that shares a row and handed to the other three by DPP (4 broadcasts per entry pair), and the
// bound test, the list loads and the pass epilogue are amortised over 4 items per lane.  The kernel is bound by that
// instruction stream together with the LDS gathers (live in the bench line: VALU ~64 %, LDS ~59 % busy, not overlapping
// fully).  With a lane that owns its row outright there is nothing to broadcast:
//     per entry pair and lane   8 ds_read_b128 (the two prev-states' whole 64-byte tile rows), 32 v_add_f32, 16 v_max3_f32,
//                               6 address adds                      = 54 instructions for 32 cells (1.69 per cell)
// against 16 for 8 cells (2.0 per cell, + test and epilogue) -- the same LDS bytes per cell.  A wave is 64 rows x 16
// items, so a pass walks as deep as the deepest of 1024 pairs (+8-10 % list blocks against 256 pairs); 23 row groups
// at 1440 states = two passes per wave and timestep on 12 waves.
//
// LDS banks.  A lane reads the whole tile row [prev-state][16 items] of an entry as four 16-byte columns.  If every lane
// took column c in instruction c, the 16 lanes of a ds_read_b128 lane group would hit 4 of the 16 bank quads (4-way
// conflict).  So lane l takes column (c + l) mod 4 in instruction c -- its accumulators are simply named in that order --
// and the four lanes of a lane group that share a column hold an ALIGNED ROW QUAD (lane_row below), which is exactly what
// arrange_blocks_kernel<4> keeps conflict-poor.  The rotation of column 0 is folded into the list offsets when the lists
// are laid out per lane (lists_by_lane_kernel); columns 1..3 cost one address add each.
//
// Lists.  A lane walks its own row, so the rows of a 64-row group are interleaved per 16-byte chunk (two entries):
// [row group][block][chunk][lane] -- one wave instruction reads 1 KB of contiguous memory.
//
// Same sorted rows, same seed (one per item), same bound, same arithmetic as resident_forward_kernel: posterior rows --
// and so the indices -- are bit-identical.  Shapes: 16-item tiles (64 <= S <= 2048), whole tiles per workgroup, one seed
// per item; everything else stays on resident_forward_kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>

#include "resident_forward.hpp"

namespace wide {

using resident::Batch;
using resident::Group;
using resident::u64;
using resident::top_key;
using resident::top_insert;

constexpr int kNI = 16;            // items per tile
constexpr int kRows = 64;          // next-states per wave pass
constexpr int kTop = 2;            // one seed + the bound
constexpr int kBlk = pruned::kBlk; // list entries per termination test
constexpr int kChunks = kBlk / 2;  // 16-byte chunks (two entries) per block and row

inline bool supported(int S) { return S >= 64 && S <= pruned::kMaxS16; }
inline int row_groups(int S) { return (S + kRows - 1) / kRows; }
// float4 elements of the per-lane lists: [row group][block][chunk][lane]
inline size_t list_chunks(int S, int SpP) { return (size_t)row_groups(S) * (SpP / kBlk) * kChunks * 64; }
// dynamic LDS: resident's layout for kTop = 2, + the bounds of the coming timestep [16] + a word of live-item bits
inline size_t lds_bytes(int S) { return resident::lds_bytes(S, kTop) + sizeof(float) * kNI + 4 * sizeof(int); }

// lane -> row among the wave's 64.  ds_read_b128 serves the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) in
// one LDS cycle each; within a group the four lanes with equal (lane & 3) read the same tile column at the same time and
// get the aligned row quad 4 * (4 * group + (lane & 3)) .. + 3.
__host__ __device__ inline int lane_row(int lane) {
    const int l5 = lane & 31;
    const bool second = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
    const int rank = second ? (l5 < 8 ? 0 : l5 < 12 ? 1 : l5 < 20 ? 2 : 3) : (l5 < 4 ? 0 : l5 < 16 ? 1 : l5 < 24 ? 2 : 3);
    const int group = 2 * (lane >> 5) + (second ? 1 : 0);
    return 16 * group + 4 * (lane & 3) + rank;
}

// once per matrix, after the sort and the arrangement: the lists per lane.  grid = (blocks per row, row groups), block = 64.
// Offsets come out rotated by the lane's first column (16 * (lane & 3) bytes into the tile row).
__global__ __launch_bounds__(64) void lists_by_lane_kernel(const float2 *__restrict__ sorted, float4 *__restrict__ lists, int S,
                                                           int SpP) {
    const int kb = blockIdx.x, rg = blockIdx.y, lane = threadIdx.x;
    const int NB = SpP / kBlk;
    int row = kRows * rg + lane_row(lane);
    row = row < S ? row : S - 1;                        // (lanes past the last state walk a valid row and store nothing)
    const float2 *src = sorted + (size_t)row * SpP + kb * kBlk;
    float4 *dst = lists + ((size_t)(rg * NB + kb) * kChunks) * 64 + lane;
    const int turn = 16 * (lane & 3);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        const float2 a = src[2 * c], b = src[2 * c + 1];
        dst[(size_t)c * 64] = make_float4(a.x, __int_as_float(__float_as_int(a.y) + turn), b.x, __int_as_float(__float_as_int(b.y) + turn));
    }
}

// ---------------------------------------------------------------------------------------
// grid = tiles of every batch of the group, block = 64 * KW, dynamic LDS = lds_bytes(S).
// MAXP >= ceil(ceil(S / 64) / KW) passes per wave and timestep.  NB = list blocks per row (SpP / 16).
// ---------------------------------------------------------------------------------------
template <int KW, int MAXP, bool PIPE>
__global__ __launch_bounds__(64 * KW) void wide_forward_kernel(Group grp, const float *__restrict__ tt,
                                                               const float4 *__restrict__ lists,
                                                               const float *__restrict__ initial, int S, int NB) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S4 = (S + 3) / 4 * 4;
    u64 *top = reinterpret_cast<u64 *>(lds + (size_t)kNI * S4);       // [16][kTop] this timestep's largest outputs
    float *mtopv = reinterpret_cast<float *>(top + kNI * kTop);       // [16][kTop] previous timestep's, decoded
    int *mtopi = reinterpret_cast<int *>(mtopv + kNI * kTop);         // their states, as offsets into tt (state * S)
    int *sframes = mtopi + kNI * kTop;                                // [16] frames per item (0 past the batch)
    int *sitem = sframes + kNI;                                       // [16] item numbers (a valid one past the batch)
    int *smisc = sitem + kNI;                                         // [0] bits of the items alive in the coming timestep
    float *sthr = reinterpret_cast<float *>(smisc + 4);               // [16] bound of the coming timestep (-inf: item ended)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (grp.only && grp.only[blockIdx.x] == 0u) return;

    const int code = grp.tile_map[blockIdx.x];
    const Batch &bat = grp.batch[code >> 20];
    const float *__restrict__ obs = bat.obs;
    float *__restrict__ hist = bat.hist;
    const int B = bat.B, T = bat.T;
    const int b0 = (code & 0xfffff) * kNI;

    if (tid < kNI) {
        int f = 0;
        const int item = bat.order[b0 + tid < B ? b0 + tid : B - 1];
        if (b0 + tid < B) {
            f = bat.frames[item];
            f = f < 1 ? 1 : (f > T ? T : f);
        }
        sframes[tid] = f;
        sitem[tid] = item;
    }
    if (tid < kNI * kTop) top[tid] = 0ull;
    __syncthreads();
    int fmax = 0;
#pragma unroll
    for (int it = 0; it < kNI; ++it) fmax = max(fmax, sframes[it]);

    // t = 0: posterior row 0 = obs[b,0,:] + initial (viterbi.cpp:72-76) into the tile, the history and the top lists
    for (int item = 0; item < kNI; ++item) {
        const int b = sitem[item];
        const bool valid = b0 + item < B;
        const float *src = obs + (size_t)b * T * S;
        float *dst = hist + (size_t)b * T * S;
        for (int i = tid; i < S; i += 64 * KW) {
            const float v = src[i] + initial[i];
            lds[i * kNI + item] = v;
            if (valid) dst[i] = v;
            top_insert<kTop>(top + item * kTop, top_key(v, i));
        }
    }

    // the largest entries of the row the tile holds -> seeds and bounds of the coming timestep `row + 1`; the running lists
    // are emptied for that timestep's outputs; the row's maximum is left for the backtrace (lazy_backtrace.hpp)
    auto publish_top = [&](int row) {
        if (tid < kNI * kTop) {
            const u64 key = top[tid];
            unsigned u = (unsigned)(key >> 32);
            u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
            const float value = key ? __uint_as_float(u) : -INFINITY;
            mtopv[tid] = value;
            mtopi[tid] = key ? (0x7fffffff - (int)(unsigned)key) * S : 0;
            top[tid] = 0ull;
            const int item = tid / kTop;
            const bool alive = row + 1 < sframes[item];
            if (tid == item * kTop && b0 + item < B && row < sframes[item]) bat.rowmax[(size_t)sitem[item] * T + row] = value;
            if (tid == item * kTop + 1) sthr[item] = alive ? value : -INFINITY;
            const unsigned long long votes = __builtin_amdgcn_ballot_w64(alive && tid == item * kTop);
            if (tid == 0) {
                unsigned bits = 0u;
#pragma unroll
                for (int it = 0; it < kNI; ++it) bits |= (unsigned)((votes >> (kTop * it)) & 1ull) << it;
                smisc[0] = (int)bits;
            }
        }
    };
    __syncthreads();
    publish_top(0);

    // this lane: row r of the wave's 64, tile columns in the order (c + turn) & 3
    const int r = lane_row(lane);
    const int turn = lane & 3;
    const char *ptile = reinterpret_cast<const char *>(lds);
    int delta[4];                      // byte offset of column (c + turn) & 3 from column turn (the lists carry column turn)
#pragma unroll
    for (int c = 0; c < 4; ++c) delta[c] = 16 * (((c + turn) & 3) - turn);
    const int nrg = (S + kRows - 1) / kRows;

    float pend[MAXP][4][4];
    unsigned stat_blocks = 0, stat_passes = 0;

    for (int t = 1; t < fmax; ++t) {
        __syncthreads();      // tile = posterior row t-1, mtop / sthr = its largest entries, `top` is empty
        const unsigned alive = (unsigned)smisc[0];

        // (an opaque zero keeps the row-group addresses of all MAXP passes from being hoisted out of the time loop)
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int rg = wave + KW * p + opaque;          // wave-uniform
            if (rg < nrg) {
                const int jj = kRows * rg + r;
                const bool jv = jj < S;
                const int jr = jv ? jj : S - 1;
                const float4 *lst = lists + ((size_t)rg * NB * kChunks) * 64 + lane;
                float4 half0[4], half1[4];               // entries 0-7 / 8-15 of the block in hand, two per register quad
#pragma unroll
                for (int c = 0; c < 4; ++c) half0[c] = lst[(size_t)c * 64];
#pragma unroll
                for (int c = 0; c < 4; ++c) half1[c] = lst[(size_t)(4 + c) * 64];
                // the seed candidates fl(seed posterior + trans[jr][seed]) start the maxima; the observations wait in `pend`
                float best[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ig = (c + turn) & 3;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int item = 4 * ig + it;
                        best[c][it] = tt[(unsigned)(mtopi[item * kTop] + jr)];       // trans[jr][i_seed]
                        pend[p][c][it] = obs[((size_t)sitem[item] * T + t) * S + jr];
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ig = (c + turn) & 3;
#pragma unroll
                    for (int it = 0; it < 4; ++it) best[c][it] = mtopv[(4 * ig + it) * kTop] + best[c][it];
                }

                struct PairData { float4 p0[4], p1[4]; float t0, t1; };
                auto issue = [&](const float4 &e, PairData &d) {       // e = {t0, offset0, t1, offset1}
                    const int o0 = __float_as_int(e.y), o1 = __float_as_int(e.w);
                    d.t0 = e.x; d.t1 = e.z;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        d.p0[c] = *reinterpret_cast<const float4 *>(ptile + (o0 + delta[c]));
                        d.p1[c] = *reinterpret_cast<const float4 *>(ptile + (o1 + delta[c]));
                    }
                };
                auto math = [&](const PairData &d) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        best[c][0] = fmaxf(fmaxf(best[c][0], d.t0 + d.p0[c].x), d.t1 + d.p1[c].x);
                        best[c][1] = fmaxf(fmaxf(best[c][1], d.t0 + d.p0[c].y), d.t1 + d.p1[c].y);
                        best[c][2] = fmaxf(fmaxf(best[c][2], d.t0 + d.p0[c].z), d.t1 + d.p1[c].z);
                        best[c][3] = fmaxf(fmaxf(best[c][3], d.t0 + d.p0[c].w), d.t1 + d.p1[c].w);
                    }
                };
                PairData ahead, other;     // PIPE: `ahead` = the pair about to be consumed, its posteriors in flight
                // the four pairs of a half block; PIPE: pair 0 is in `ahead` already, the next half's pair 0 is left there
                auto consume = [&](const float4 (&h)[4], const float4 &after) {
                    if constexpr (PIPE) {
                        issue(h[1], other);
                        __builtin_amdgcn_sched_barrier(0);
                        math(ahead);
                        __builtin_amdgcn_sched_barrier(0);
                        issue(h[2], ahead);
                        __builtin_amdgcn_sched_barrier(0);
                        math(other);
                        __builtin_amdgcn_sched_barrier(0);
                        issue(h[3], other);
                        __builtin_amdgcn_sched_barrier(0);
                        math(ahead);
                        __builtin_amdgcn_sched_barrier(0);
                        issue(after, ahead);
                        __builtin_amdgcn_sched_barrier(0);
                        math(other);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            issue(h[k], ahead);
                            math(ahead);
                        }
                    }
                };
                // any (row, item) pair of this wave whose bound still exceeds its maximum?
                auto more = [&](float tn) {
                    bool open = false;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float4 th = *reinterpret_cast<const float4 *>(sthr + 4 * ((c + turn) & 3));
                        open = open | (tn + th.x > best[c][0]) | (tn + th.y > best[c][1]) | (tn + th.z > best[c][2]) |
                               (tn + th.w > best[c][3]);
                    }
                    return (bool)__any(jv && open);
                };

                if constexpr (PIPE) issue(half0[0], ahead);
                int nblk = 0;                              // wave-uniform: blocks consumed
                const int Sp = (S + 15) / 16 * 16;
                for (int kb = 0; kb * kBlk < Sp; ++kb) {
                    ++nblk;
                    consume(half0, half1[0]);
                    const float4 *nxt = lst + (size_t)(kb + 1) * kChunks * 64;      // (kPad blocks of -inf follow every row)
#pragma unroll
                    for (int c = 0; c < 4; ++c) half0[c] = nxt[(size_t)c * 64];
                    consume(half1, half0[0]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) half1[c] = nxt[(size_t)(4 + c) * 64];
                    if (!more(half0[0].x)) break;
                }
                if ((t & 15) == 1) { stat_blocks += (unsigned)nblk; stat_passes += 1u; }

                // outputs: post'[j] = obs[t,j] + max (viterbi.cpp:102); history, running top lists
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ig = (c + turn) & 3;
                    u64 last[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) last[it] = top[(4 * ig + it) * kTop + kTop - 1];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int item = 4 * ig + it;
                        const float o = pend[p][c][it] + best[c][it];
                        pend[p][c][it] = o;
                        if (jv && ((alive >> item) & 1u)) hist[((size_t)sitem[item] * T + t) * S + jr] = o;
                        const u64 key = top_key(o, jr);
                        if (jv && key > last[it]) top_insert<kTop>(top + item * kTop, key);
                    }
                }
            }
        }
        __syncthreads();      // every wave is done reading the tile, mtop and sthr; every output is in `top`
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int rg = wave + KW * p + opaque;
            const int jj = kRows * rg + r;
            if (rg < nrg && jj < S) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *reinterpret_cast<float4 *>(lds + (size_t)jj * kNI + 4 * ((c + turn) & 3)) =
                        make_float4(pend[p][c][0], pend[p][c][1], pend[p][c][2], pend[p][c][3]);
            }
        }
        publish_top(t);
    }
    if (lane == 0 && stat_passes) {
        // (statistics in units of resident_forward_kernel's: list blocks per wave pass)
        atomicAdd(&grp.stats[0], stat_blocks);
        atomicAdd(&grp.stats[64], stat_passes);
    }
}

}  // namespace wide
