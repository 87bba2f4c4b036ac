// small_batch_forward.hpp -- exact forward recurrence for SMALL batches (B <= 16), one launch per timestep.
//
// The generic row kernels (torbi_hip.hip, step_rows*_kernel) stream the whole S x S transition matrix through the
// chip for every timestep of every item -- 8.3 MB at S = 1440: 5.2 us per launch at B = 1 (rocprofv3), the matrix
// does not fit the 4 MB L2s.  Here every wave owns ONE next-state j of ONE item and walks row j's SORTED list
// (pruned_forward.hpp: sort_rows_kernel) in chunks of 64 entries, one entry per lane, against the item's posterior row
// staged in the LDS: candidate = fl(post[i_k] + t_k); every entry not yet examined has t <= t_next (the next chunk's
// first entry) and post <= pmax, so once fl(t_next + pmax) <= best the maximum is final (monotone rounding; values
// only -- the backpointer is recomputed along the decoded path by lazy_backtrace.hpp).  With one item per wave there
// is no lock-step across items: a row needs one chunk on the benchmark inputs (52 entries on average), i.e. 0.5 KB of
// list instead of a 5.76 KB transition row.  Posterior rows are bit-identical to the reference's (viterbi.cpp:78-108).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "wave_reduce.hpp"

namespace rowscan {

constexpr int kRowsPerBlock = 4;      // one wave per next-state, four per workgroup

inline bool supported(int B, int S) { return B >= 1 && B <= 16 && S >= 64 && S <= 4096; }

// Where it beats the generic row kernels (tools/small_batch_probe.py, T = 500).  Both are bound by the ~4.4 us between
// dependent launches at B = 1, S = 1440 (rows 2.20 + 0.53 ms backtrace over the sorted rows against 2.30 + 0.13 ms trellis
// chase); the list walk wins once a launch carries real work: B = 16, S = 1440: 6.0 vs 6.8 ms; B = 1, S = 4096: 5.8 vs
// 7.5 ms; B = 8, S = 4096: 14.9 vs 28.1 ms.  AUTO takes the kernel there; TORBI_HIP_FORWARD_PRUNED names it for any B <= 16.
// (round 3, with the gather backtrace -- 0.39 instead of 0.61 ms -- and the parallel chase behind the generic kernels:
// 6 x 500 x 1440 3.55 against 3.99 ms, 4 x 500 x 1440 2.98 against 2.76: profiles/r03_held_probe.txt)
inline bool profitable(int B, int S) { return supported(B, S) && (S > 2048 || B >= 6); }

// One timestep of every item.  grid = (ceil(S / 4), B), block = 256, dynamic LDS = 4 * S bytes.  Leaves the maximum of
// posterior row t-1 in rowmax[b][t-1].
// List entries are {t, prev-state << shift} (sort_rows_kernel with row_bytes = 1 << shift).
__global__ __launch_bounds__(256) void step_rows_sorted_kernel(const float *__restrict__ obs,
                                                               const int32_t *__restrict__ frames,
                                                               const float2 *__restrict__ sorted, float *__restrict__ hist,
                                                               float *__restrict__ rowmax, int B, int T, int S, int t,
                                                               int SpP, int shift) {
    extern __shared__ __attribute__((aligned(16))) float prow[];
    __shared__ float xmax[kRowsPerBlock];
    const int b = blockIdx.y;
    int f = frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    if (t >= f) return;                                       // block-uniform: the item has ended
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = blockIdx.x * kRowsPerBlock + wave;
    const int jr = j < S ? j : S - 1;
    // the first list chunk and the observation do not depend on the posterior: request them first
    const float2 *row = sorted + (size_t)jr * SpP;
    float2 ent = row[lane];
    const float *item = hist + (size_t)b * T * S;
    const float ob = obs[((size_t)b * T + t) * S + jr];
    // posterior row t-1 -> LDS, and its maximum
    const float *prev = item + (size_t)(t - 1) * S;
    float lm = -INFINITY;
    for (int i = tid; i < S; i += 256) {
        const float v = prev[i];
        prow[i] = v;
        lm = __builtin_fmaxf(lm, v);
    }
    const float wm = wavered::wave_reduce_f32(lm, wavered::MaxOp());
    if (lane == 0) xmax[wave] = wm;
    __syncthreads();
    const float pmax = __builtin_fmaxf(__builtin_fmaxf(xmax[0], xmax[1]), __builtin_fmaxf(xmax[2], xmax[3]));
    if (blockIdx.x == 0 && tid == 0) rowmax[(size_t)b * T + t - 1] = pmax;        // for the backtrace's bound (lazy_backtrace.hpp)

    const int Sp = (S + 15) / 16 * 16;
    float bv = -INFINITY, best = -INFINITY;
    for (int k0 = 0; k0 < Sp; k0 += 64) {
        const int kn = k0 + 64 + lane;
        const float2 ahead = row[kn < SpP ? kn : SpP - 1];        // next chunk (its first entry bounds the rest)
        if (k0 + lane < Sp) bv = __builtin_fmaxf(bv, prow[__float_as_int(ent.y) >> shift] + ent.x);
        best = wavered::wave_reduce_f32(bv, wavered::MaxOp());
        const float tn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__float_as_int(ahead.x)));
        ent = ahead;
        if (k0 + 64 >= Sp || tn + pmax <= best) break;
    }
    if (lane == 0 && j < S) hist[((size_t)b * T + t) * S + j] = ob + best;          // post'[j] = obs[t,j] + max
}

}  // namespace rowscan
