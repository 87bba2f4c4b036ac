// lazy_backtrace.hpp -- final state, tail fill and backtrace for the dense forward path.
//
// The dense forward pass (dense_forward.hpp) keeps the posterior rows hist[b][t][:] instead of a
// backpointer trellis.  The backtrace needs exactly one backpointer per (b, t): the one of the
// state on the decoded path.  It is recomputed here with the reference's own arithmetic,
//     bp = first argmax_i fl( hist[b][t-1][i] + trans[j][i] )          (viterbi.cpp:81-100)
// so decoded indices are identical to materialising the whole trellis (viterbi.cpp:153-157).
// One wave per batch item; each lane scans an ascending subsequence of i with the reference's
// strict '>' and the wave reduction keeps the lowest index among equal maxima.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "wave_reduce.hpp"

namespace lazy {

constexpr int kSentinel = 0x7fffffff;

__device__ __forceinline__ void take_better(float &v, int &i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__device__ __forceinline__ int wave_first_argmax(float v, int i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        take_better(v, i, ov, oi);
    }
    return i;   // every lane holds the result (xor butterfly)
}

__device__ __forceinline__ void scan1(float c, int i, float &best, int &arg) {
    if (c > best) { best = c; arg = i; }
    else if (arg == kSentinel) { best = c; arg = i; }   // first candidate, also when it is -inf
}

// VEC = 4: S % 4 == 0, rows 16-byte aligned -> float4 loads; VEC = 1: any S
template <int VEC>
__device__ __forceinline__ void backtrace_item(const float *__restrict__ h, const float *__restrict__ trans, int f,
                                               int32_t *__restrict__ o, int T, int S, int lane) {
    f = f < 1 ? 1 : (f > T ? T : f);

    // final state = first argmax of the last posterior row (viterbi.cpp:218)
    float best = -INFINITY;
    int arg = kSentinel;
    {
        const float *row = h + (size_t)(f - 1) * S;
        if (VEC == 4) {
            for (int i = 4 * lane; i < S; i += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(row + i);
                scan1(v.x, i, best, arg); scan1(v.y, i + 1, best, arg);
                scan1(v.z, i + 2, best, arg); scan1(v.w, i + 3, best, arg);
            }
        } else {
            for (int i = lane; i < S; i += 64) scan1(row[i], i, best, arg);
        }
    }
    int j = wave_first_argmax(best, arg);

    // every position t >= frames-1 holds the final state (viterbi.cpp:219-221)
    for (int tt = f - 1 + lane; tt < T; tt += 64) o[tt] = j;

    for (int tt = f - 1; tt >= 1; --tt) {
        const float *row = h + (size_t)(tt - 1) * S;
        const float *tr = trans + (size_t)j * S;
        best = -INFINITY;
        arg = kSentinel;
        if (VEC == 4) {
            for (int i = 4 * lane; i < S; i += 256) {
                const float4 p = *reinterpret_cast<const float4 *>(row + i);
                const float4 q = *reinterpret_cast<const float4 *>(tr + i);
                scan1(p.x + q.x, i, best, arg); scan1(p.y + q.y, i + 1, best, arg);
                scan1(p.z + q.z, i + 2, best, arg); scan1(p.w + q.w, i + 3, best, arg);
            }
        } else {
            for (int i = lane; i < S; i += 64) scan1(row[i] + tr[i], i, best, arg);
        }
        j = wave_first_argmax(best, arg);
        if (lane == 0) o[tt - 1] = j;
    }
}

template <int VEC>
__global__ __launch_bounds__(64) void backtrace_kernel(const float *__restrict__ hist,
                                                       const float *__restrict__ trans,
                                                       const int32_t *__restrict__ frames,
                                                       int32_t *__restrict__ out, int B, int T, int S) {
    const int b = blockIdx.x;
    backtrace_item<VEC>(hist + (size_t)b * T * S, trans, frames[b], out + (size_t)b * T, T, S, threadIdx.x);
}

// first index (ascending) among this lane's 4*NQ elements whose value equals m, else kSentinel
template <int NQ>
__device__ __forceinline__ int lane_first_equal(const float4 (&v)[NQ], float m, int lane, int S) {
    int k = kSentinel;
#pragma unroll
    for (int q = NQ - 1; q >= 0; --q) {
        const int i = 4 * lane + 256 * q;
        if (i < S) {
            int kq = v[q].w == m ? i + 3 : kSentinel;
            kq = v[q].z == m ? i + 2 : kq;
            kq = v[q].y == m ? i + 1 : kq;
            kq = v[q].x == m ? i : kq;
            k = min(k, kq);
        }
    }
    return k;
}

template <int NQ>
__device__ __forceinline__ float lane_max(const float4 (&v)[NQ], int lane, int S) {
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (4 * lane + 256 * q < S)
            m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(v[q].x, v[q].y)), __builtin_fmaxf(v[q].z, v[q].w));
    return m;
}

// first argmax over the wave: max value by DPP all-reduce (exact, order independent), then the
// lowest index whose candidate equals it -- the reference's strict-'>' scan (viterbi.cpp:94-100)
template <int NQ>
__device__ __forceinline__ int wave_first_argmax4(const float4 (&v)[NQ], int lane, int S) {
    const float m = wavered::wave_reduce_f32(lane_max<NQ>(v, lane, S), wavered::MaxOp());
    const int k = wavered::wave_min_i32(lane_first_equal<NQ>(v, m, lane, S));
    return k < S ? k : 0;          // (a row of NaNs equals nothing: stay inside the matrix -- nonfinite.hpp decodes the item again)
}

// Register-resident form for S % 4 == 0 and S <= 256*NQ: the posterior row of the NEXT path step
// does not depend on the state being resolved, so it is prefetched into registers while the current
// step's transition row (which does) is in flight; one wave per batch item, lanes own 4 consecutive
// prev-states per 256-wide stripe.
template <int NQ>
__device__ __forceinline__ void backtrace_prefetch_item(const float *__restrict__ h, const float *__restrict__ trans,
                                                        int f, int32_t *__restrict__ o, int T, int S, int lane) {
    f = f < 1 ? 1 : (f > T ? T : f);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 cur[NQ], nxt[NQ];
    {
        const float *row = h + (size_t)(f - 1) * S;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            cur[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : zero;
        }
    }
    {   // row f-2 (clamped to row 0: valid memory, unused when f == 1)
        const float *row = h + (size_t)(f >= 2 ? f - 2 : 0) * S;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            nxt[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : zero;
        }
    }
    // final state = first argmax of the last posterior row (viterbi.cpp:218)
    int j = wave_first_argmax4<NQ>(cur, lane, S);
    // every position t >= frames-1 holds the final state (viterbi.cpp:219-221)
    for (int tt = f - 1 + lane; tt < T; tt += 64) o[tt] = j;

    for (int tt = f - 1; tt >= 1; --tt) {
        // transition row of the state just resolved (depends on j) ...
        const float *tr = trans + (size_t)j * S;
        float4 q4[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            q4[q] = i < S ? *reinterpret_cast<const float4 *>(tr + i) : zero;
        }
        // ... posterior row tt-1 is already in registers; fetch row tt-2 for the next step
#pragma unroll
        for (int q = 0; q < NQ; ++q) cur[q] = nxt[q];
        {
            const float *row = h + (size_t)(tt >= 2 ? tt - 2 : 0) * S;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int i = 4 * lane + 256 * q;
                nxt[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : zero;
            }
        }
        float4 cand[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            cand[q] = make_float4(cur[q].x + q4[q].x, cur[q].y + q4[q].y, cur[q].z + q4[q].z, cur[q].w + q4[q].w);
        j = wave_first_argmax4<NQ>(cand, lane, S);
        if (lane == 0) o[tt - 1] = j;
    }
}

// (`narrow`: null, or the widest row window of the matrix as row_ranges_kernel left it -- at most kRangedWindow means
// backtrace_ranged_kernel decodes this launch and this kernel has nothing to do)
template <int NQ>
__global__ __launch_bounds__(64) void backtrace_prefetch_kernel(const float *__restrict__ hist,
                                                                const float *__restrict__ trans,
                                                                const int32_t *__restrict__ frames,
                                                                int32_t *__restrict__ out, int B, int T, int S,
                                                                const int32_t *__restrict__ narrow) {
    if (narrow && *narrow <= 512) return;
    const int b = blockIdx.x;
    backtrace_prefetch_item<NQ>(hist + (size_t)b * T * S, trans, frames[b], out + (size_t)b * T, T, S, threadIdx.x);
}

// ---------------------------------------------------------------------------------------
// The same backtrace where the forward pass left SORTED transition rows (pruned / time-resident paths): per path
// step the argmax_i fl(hist[t-1][i] + trans[j][i]) is found by walking row j in descending transition order, 64
// entries per wave step, against the posterior row held in the LDS:
//     every entry not yet examined has trans <= t_next (the next chunk's first entry) and hist <= hmax, so its
//     candidate is <= fl(t_next + hmax); once that is < the best candidate seen, no unexamined entry can reach --
//     or TIE -- the maximum, and the lowest prev-state among the examined maxima is the reference's backpointer
//     (viterbi.cpp:94-100).
// A step then moves the 4S-byte posterior row (prefetched one step ahead, coalesced) plus 0.5-1 KB of list instead
// of the posterior row plus a 4S-byte transition row: with 4096 items in flight (a launch group of 8 batches) the
// backtrace is bound by exactly that traffic.  List entries are {t, prev-state << shift}; (-inf) entries and the
// row padding carry a stand-in prev-state, which matters only when EVERY candidate is -inf: the reference's scan
// then keeps prev-state 0, and so does this.
// One wave per item; dynamic LDS = 4 * ceil4(S) bytes per wave; S % 4 == 0, S <= 256 * NQ.
// ---------------------------------------------------------------------------------------
template <int NQ>
__device__ __forceinline__ void backtrace_sorted_item(const float *__restrict__ h, const float2 *__restrict__ sorted,
                                                      int SpP, int shift, int f, int32_t *__restrict__ o, int T, int S,
                                                      int lane, float *__restrict__ hrow) {
    f = f < 1 ? 1 : (f > T ? T : f);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nxt[NQ];
    auto load_row = [&](float4 (&dst)[NQ], int r) {
        const float *row = h + (size_t)r * S;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            dst[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : zero;
        }
    };
    load_row(nxt, f - 1);
    // final state = first argmax of the last posterior row (viterbi.cpp:218)
    int j = wave_first_argmax4<NQ>(nxt, lane, S);
    asm volatile("" : "+v"(j));                              // (the row below is requested behind the argmax: one row live)
    load_row(nxt, f >= 2 ? f - 2 : 0);
    // every position t >= frames-1 holds the final state (viterbi.cpp:219-221)
    for (int tt = f - 1 + lane; tt < T; tt += 64) o[tt] = j;
    const int Sp = (S + 15) / 16 * 16;

    for (int tt = f - 1; tt >= 1; --tt) {
        // the list of the state just resolved (depends on j): first chunk on its way ...
        const float2 *row = sorted + (size_t)j * SpP;
        float2 ent = row[lane];
        // ... while posterior row tt-1 (already in registers) goes to the LDS; row tt-2 is requested behind it, into the
        // same registers (one row live, not two: 42 registers instead of 61, so the kernel fits beside the three
        // 152-register forward waves per SIMD of the next launch group; DESIGN.md 4.12 for what that is worth)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            if (i < S) *reinterpret_cast<float4 *>(hrow + i) = nxt[q];
        }
        const float hmax = wavered::wave_reduce_f32(lane_max<NQ>(nxt, lane, S), wavered::MaxOp());
        load_row(nxt, tt >= 2 ? tt - 2 : 0);
        float bv = -INFINITY;
        int bi = kSentinel;
        float best = -INFINITY;
        for (int k0 = 0; k0 < Sp; k0 += 64) {
            const int kn = k0 + 64 + lane;
            const float2 ahead = row[kn < SpP ? kn : SpP - 1];       // next chunk (its first entry bounds the rest)
            if (k0 + lane < Sp) {
                const int i = __float_as_int(ent.y) >> shift;
                const float c = hrow[i] + ent.x;
                if (c > bv || (c == bv && i < bi)) { bv = c; bi = i; }
            }
            best = wavered::wave_reduce_f32(bv, wavered::MaxOp());
            const float tn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__float_as_int(ahead.x)));
            ent = ahead;
            if (k0 + 64 >= Sp || tn + hmax < best) break;
        }
        // lowest prev-state among the lanes that hold the maximum; all candidates -inf: prev-state 0
        const int cand = (bv == best && bi != kSentinel) ? bi : kSentinel;
        const int win = wavered::wave_min_i32(cand);
        j = best == -INFINITY ? 0 : win;
        if (lane == 0) o[tt - 1] = j;
    }
}

// The same walk without staging posterior rows: the posteriors the list chunk points at are gathered straight from the
// history (64 four-byte reads touch ~46 of a 1440-state row's 90 sectors; staging reads all of them), and the row maximum
// the bound needs comes from `rowmax`, which the time-resident forward kernel leaves behind for every row.  A step is two
// dependent loads (list chunk, then posteriors) instead of one -- and still the faster form from one 512-item batch (0.58
// against 0.80 ms) to a launch group of eight (1.86 against 3.11 ms: 4096 paths move 11.8 GB through row staging).
template <int NQ>
struct GatherWalker {
    const float *__restrict__ h;          // [T][S] posterior rows of the item
    const float *__restrict__ rowmax;     // [T] their maxima
    const float2 *__restrict__ sorted;
    int SpP, shift, S, lane;
    // first argmax of posterior row t (the final state when t = frames - 1: viterbi.cpp:218)
    __device__ __forceinline__ int first_state(int t) const {
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 last[NQ];
        const float *row = h + (size_t)t * S;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            last[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : zero;
        }
        return wave_first_argmax4<NQ>(last, lane, S);
    }
    // the state at timestep tt - 1 of the path that is in state j at timestep tt
    __device__ __forceinline__ int step(int j, int tt) const {
        const int Sp = (S + 15) / 16 * 16;
        const float2 *row = sorted + (size_t)j * SpP;
        float2 ent = row[lane];
        const float hmax = rowmax[tt - 1];                   // (independent of the path: in flight with the list chunk)
        const float *hrow = h + (size_t)(tt - 1) * S;
        float bv = -INFINITY;
        int bi = kSentinel;
        float best = -INFINITY;
        for (int k0 = 0; k0 < Sp; k0 += 64) {
            // only the FIRST entry of the next chunk is needed to bound the rest of the list (one 8-byte read for the wave
            // instead of a 512-byte chunk that one step in ten goes on to use)
            const int kn = k0 + 64 < SpP ? k0 + 64 : SpP - 1;
            const float tn = row[kn].x;
            if (k0 + lane < Sp) {
                const int i = __float_as_int(ent.y) >> shift;
                const float c = hrow[i] + ent.x;
                if (c > bv || (c == bv && i < bi)) { bv = c; bi = i; }
            }
            best = wavered::wave_reduce_f32(bv, wavered::MaxOp());
            if (k0 + 64 >= Sp || tn + hmax < best) break;
            const int kl = k0 + 64 + lane;
            ent = row[kl < SpP ? kl : SpP - 1];
        }
        const int cand = (bv == best && bi != kSentinel) ? bi : kSentinel;
        const int win = wavered::wave_min_i32(cand);
        return best == -INFINITY ? 0 : win;
    }
};

// (A row of NaNs has no first argmax: the walkers then answer kSentinel.  Such an item is decoded again behind this launch
// (nonfinite.hpp); here the state only has to stay inside the matrix.)
template <class Walker>
__device__ __forceinline__ int inside(const Walker &w, int j) { return (unsigned)j < (unsigned)w.S ? j : 0; }

// final state, tail fill (viterbi.cpp:218-221) and the walk down the whole path: one wave per item
template <class Walker>
__device__ __forceinline__ void walk_item(const Walker &w, int f, int32_t *__restrict__ o, int T, int lane) {
    f = f < 1 ? 1 : (f > T ? T : f);
    int j = inside(w, w.first_state(f - 1));
    for (int tt = f - 1 + lane; tt < T; tt += 64) o[tt] = j;
    for (int tt = f - 1; tt >= 1; --tt) {
        j = inside(w, w.step(j, tt));
        if (lane == 0) o[tt - 1] = j;
    }
}

// ---- the same walk in K SPECULATIVE SEGMENTS (single batches: a path per wave leaves the chip idle and pays two dependent
// loads per step, 0.6 ms for 500 steps).  Backward walks from different states of a timestep merge with the decoded path
// within a few steps (0-22 on the benchmark's rows and on posteriorgram-like rows under the pitch band), so segment s of an
// item -- timesteps (lo, hi], lo = s L, hi = (s + 1) L, L = ceil((f - 1) / K) -- is walked by a wave of its own from the
// first argmax of posterior row hi, all K at once (chase_segment), and one wave per item then checks the joints from the
// end of the path (stitch_segments): where a segment's start is not the state the path really is in at hi, it walks on from
// the true state until it meets what the segment left -- from there down the segment's walk IS the path -- or the segment's
// end.  Every step is the same arithmetic; the result is the walk_item path, state for state.  `arrive[s]`: the state
// segment s reached at timestep lo.
template <class Walker>
__device__ __forceinline__ void chase_segment(const Walker &w, int f, int T, int K, int s, int32_t *__restrict__ o,
                                              int32_t *__restrict__ arrive, int lane) {
    f = f < 1 ? 1 : (f > T ? T : f);
    const int L = (f - 1 + K - 1) / K;
    const int lo = min(s * L, f - 1), hi = min((s + 1) * L, f - 1);
    const bool last = s == K - 1;                         // (hi = f - 1: the one segment whose start is known)
    if (!last && lo >= hi) return;
    int j = inside(w, w.first_state(hi));
    if (last) {
        for (int tt = f - 1 + lane; tt < T; tt += 64) o[tt] = j;
    } else if (lane == 0) {
        o[hi] = j;
    }
    for (int tt = hi; tt > lo; --tt) {
        j = inside(w, w.step(j, tt));
        if (lane == 0 && (tt - 1 > lo || s == 0)) o[tt - 1] = j;      // (o[lo] is the start of segment s - 1)
    }
    if (lane == 0) arrive[s] = j;
}

// (`stats`: null, or two counters -- steps walked, steps walked AGAIN at the joints: what the speculation cost)
template <class Walker>
__device__ __forceinline__ void stitch_segments(const Walker &w, int f, int T, int K, int32_t *__restrict__ o,
                                                const int32_t *__restrict__ arrive, int lane, unsigned *stats = nullptr) {
    f = f < 1 ? 1 : (f > T ? T : f);
    const int L = (f - 1 + K - 1) / K;
    int truth = arrive[K - 1];                            // the path's state at the last segment's lo
    unsigned again = 0;
    for (int s = K - 2; s >= 0; --s) {
        const int lo = min(s * L, f - 1), hi = min((s + 1) * L, f - 1);
        if (lo >= hi) continue;
        if (o[hi] == truth) {                             // the segment started where the path is: all of it stands
            truth = arrive[s];
            continue;
        }
        int j = truth;
        if (lane == 0) o[hi] = j;
        bool met = false;
        for (int tt = hi; tt > lo; --tt) {
            j = inside(w, w.step(j, tt));
            ++again;
            if (tt - 1 == lo) break;
            if (o[tt - 1] == j) { met = true; break; }
            if (lane == 0) o[tt - 1] = j;
        }
        if (met) {
            truth = arrive[s];
        } else {
            truth = j;
            if (s == 0 && lane == 0) o[0] = j;
        }
    }
    if (stats && lane == 0) {
        atomicAdd(stats, (unsigned)(f - 1));
        atomicAdd(stats + 1, again);
    }
}

template <int NQ>
__device__ __forceinline__ void backtrace_gather_item(const float *__restrict__ h, const float *__restrict__ rowmax,
                                                      const float2 *__restrict__ sorted, int SpP, int shift, int f,
                                                      int32_t *__restrict__ o, int T, int S, int lane) {
    const GatherWalker<NQ> w{h, rowmax, sorted, SpP, shift, S, lane};
    walk_item(w, f, o, T, lane);
}

template <int NQ>
__global__ __launch_bounds__(64) void backtrace_sorted_kernel(const float *__restrict__ hist,
                                                              const float2 *__restrict__ sorted, int SpP, int shift,
                                                              const int32_t *__restrict__ frames,
                                                              int32_t *__restrict__ out, int B, int T, int S) {
    extern __shared__ __attribute__((aligned(16))) float hrow_lds[];
    const int b = blockIdx.x;
    backtrace_sorted_item<NQ>(hist + (size_t)b * T * S, sorted, SpP, shift, frames[b], out + (size_t)b * T, T, S,
                              threadIdx.x, hrow_lds);
}

template <int NQ>
__global__ __launch_bounds__(64) void backtrace_gather_kernel(const float *__restrict__ hist, const float *__restrict__ rowmax,
                                                              const float2 *__restrict__ sorted, int SpP, int shift,
                                                              const int32_t *__restrict__ frames, int32_t *__restrict__ out,
                                                              int B, int T, int S) {
    const int b = blockIdx.x;
    backtrace_gather_item<NQ>(hist + (size_t)b * T * S, rowmax + (size_t)b * T, sorted, SpP, shift, frames[b],
                              out + (size_t)b * T, T, S, threadIdx.x);
}

// ---------------------------------------------------------------------------------------
// Banded matrices (the dense route's usual guest: the reference's pitch model reaches 87 states either way): outside a
// row's finite range [lo, hi) every candidate is -inf, so the path step reads the range of the transition row and of
// the posterior row -- 2 x 0.7 KB instead of 2 x 5.76 KB at 1440 states -- and not the rest.  `ranges` holds {lo, hi} per
// row and `widest` the largest hi - lo4 of the matrix (row_ranges_kernel below); the kernel serves matrices whose widest
// row fits its window of 512 prev-states and returns at once otherwise (the whole-row kernel, launched next to it, then
// runs: it returns at once in the opposite case).  All candidates -inf -> prev-state 0, like the reference's scan.
// ---------------------------------------------------------------------------------------
constexpr int kRangedWindow = 512;

// grid = S, block = 64: finite range of every transition row; *widest = max over rows of the window its range needs
// (hi - (lo rounded down to 4)); zeroed by the caller
__global__ __launch_bounds__(64) void row_ranges_kernel(const float *__restrict__ trans, int32_t *__restrict__ ranges,
                                                        int32_t *__restrict__ widest, int S) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const float *row = trans + (size_t)j * S;
    int lo = kSentinel, hi = 0;
    for (int i = lane; i < S; i += 64) {
        if (row[i] != -INFINITY) {
            lo = min(lo, i);
            hi = max(hi, i + 1);
        }
    }
    lo = wavered::wave_min_i32(lo);
    hi = -wavered::wave_min_i32(-hi);
    if (lane == 0) {
        if (hi == 0) lo = 0;                                     // nothing finite: an empty range
        ranges[2 * j] = lo;
        ranges[2 * j + 1] = hi;
        atomicMax(widest, hi - (lo & ~3));
    }
}

template <int NQ>
__global__ __launch_bounds__(64) void backtrace_ranged_kernel(const float *__restrict__ hist, const float *__restrict__ trans,
                                                              const int32_t *__restrict__ ranges,
                                                              const int32_t *__restrict__ widest,
                                                              const int32_t *__restrict__ frames, int32_t *__restrict__ out,
                                                              int B, int T, int S) {
    if (*widest > kRangedWindow) return;                         // a wide matrix: the whole-row kernel decodes it
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *h = hist + (size_t)b * T * S;
    int32_t *o = out + (size_t)b * T;
    int f = frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    int j;
    {
        float4 last[NQ];
        const float *row = h + (size_t)(f - 1) * S;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            last[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : zero;
        }
        j = wave_first_argmax4<NQ>(last, lane, S);
    }
    for (int tt = f - 1 + lane; tt < T; tt += 64) o[tt] = j;
    const float4 none = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int tt = f - 1; tt >= 1; --tt) {
        const int lo4 = ranges[2 * j] & ~3, hi = ranges[2 * j + 1];
        const float *tr = trans + (size_t)j * S, *hrow = h + (size_t)(tt - 1) * S;
        float4 cand[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = lo4 + 4 * lane + 256 * q;
            if (i < hi) {                                        // (i % 4 == 0 and S % 4 == 0: the four reads stay inside the row)
                const float4 t4 = *reinterpret_cast<const float4 *>(tr + i);
                const float4 p4 = *reinterpret_cast<const float4 *>(hrow + i);
                cand[q] = make_float4(p4.x + t4.x, p4.y + t4.y, p4.z + t4.z, p4.w + t4.w);
            } else {
                cand[q] = none;
            }
        }
        // first argmax inside the window (indices relative to lo4), as wave_first_argmax4 finds it over a whole row
        const float m = wavered::wave_reduce_f32(
            __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(cand[0].x, cand[0].y), __builtin_fmaxf(cand[0].z, cand[0].w)),
                            __builtin_fmaxf(__builtin_fmaxf(cand[1].x, cand[1].y), __builtin_fmaxf(cand[1].z, cand[1].w))),
            wavered::MaxOp());
        int k = kSentinel;
#pragma unroll
        for (int q = 1; q >= 0; --q) {
            const int i = lo4 + 4 * lane + 256 * q;
            int kq = cand[q].w == m ? i + 3 : kSentinel;
            kq = cand[q].z == m ? i + 2 : kq;
            kq = cand[q].y == m ? i + 1 : kq;
            kq = cand[q].x == m ? i : kq;
            k = min(k, kq);
        }
        k = wavered::wave_min_i32(k);
        j = (m == -INFINITY || k >= S) ? 0 : k;                 // (every candidate -inf: the reference's scan keeps prev-state 0;
                                                                // a window of NaNs: inside the matrix, nonfinite.hpp decodes the item again)
        if (lane == 0) o[tt - 1] = j;
    }
}

// ---- whole transition rows in speculative segments (chase_segment / stitch_segments above): the backtrace behind the
// value-only workgroup kernel of small_states.hpp (65 .. 256 states: a step reads 2 x 4 S bytes and is bound by their
// latency, not by bytes -- at 1440 states the row walks of the dense route gained 3 % and keep whole paths).  S % 4 == 0,
// 16-byte aligned matrix, S <= 256 NQ.
template <int NQ>
struct RowWalker {
    const float *__restrict__ h;          // [T][S] posterior rows of the item
    const float *__restrict__ trans;
    int S, lane;
    __device__ __forceinline__ int first_state(int t) const {
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 last[NQ];
        const float *row = h + (size_t)t * S;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            last[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : zero;
        }
        return wave_first_argmax4<NQ>(last, lane, S);
    }
    __device__ __forceinline__ int step(int j, int tt) const {
        const float *tr = trans + (size_t)j * S, *hrow = h + (size_t)(tt - 1) * S;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 cand[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            const float4 t4 = i < S ? *reinterpret_cast<const float4 *>(tr + i) : zero;
            const float4 p4 = i < S ? *reinterpret_cast<const float4 *>(hrow + i) : zero;
            cand[q] = make_float4(p4.x + t4.x, p4.y + t4.y, p4.z + t4.z, p4.w + t4.w);
        }
        return wave_first_argmax4<NQ>(cand, lane, S);
    }
};

// grid = B x K
template <int NQ>
__global__ __launch_bounds__(64) void segment_rows_kernel(const float *__restrict__ hist, const float *__restrict__ trans,
                                                          const int32_t *__restrict__ frames, int32_t *__restrict__ out, int B,
                                                          int T, int S, int K, int32_t *__restrict__ arrive) {
    const int b = (int)blockIdx.x / K, seg = (int)blockIdx.x - b * K;
    const RowWalker<NQ> w{hist + (size_t)b * T * S, trans, S, (int)threadIdx.x};
    chase_segment(w, frames[b], T, K, seg, out + (size_t)b * T, arrive + (size_t)b * K, threadIdx.x);
}
// grid = B
template <int NQ>
__global__ __launch_bounds__(64) void stitch_rows_kernel(const float *__restrict__ hist, const float *__restrict__ trans,
                                                         const int32_t *__restrict__ frames, int32_t *__restrict__ out, int B,
                                                         int T, int S, int K, const int32_t *__restrict__ arrive) {
    const int b = blockIdx.x;
    const RowWalker<NQ> w{hist + (size_t)b * T * S, trans, S, (int)threadIdx.x};
    stitch_segments(w, frames[b], T, K, out + (size_t)b * T, arrive + (size_t)b * K, threadIdx.x);
}

}  // namespace lazy
