// uniform_decode.hpp -- exact decode for a UNIFORM transition matrix (every entry = c).
//
// This is the reference's default: from_probabilities(observation) with transition=None builds
// torch.full((S, S), log(1/S)) (torbi/core.py:175-180) and runs the generic S*S recurrence on it.
// With all rows identical the candidates fl(post[i] + c) do not depend on the next state j, so per
// timestep (viterbi.cpp:78-108)
//     m    = max_i fl(post[i] + c)            k = first i attaining it     (the backpointer of EVERY j)
//     post'[j] = fl(obs[t,j] + m)
// and the backtrace (viterbi.cpp:153-157) reads bp[t][anything] = k_t:  out[t-1] = k_t.
// O(S) per timestep instead of O(S*S), no trellis, no posterior history: the decode streams the
// observations once and is HBM-bound (4S + 4 bytes per timestep).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "wave_reduce.hpp"
#include "nonfinite.hpp"

namespace uniform {

using wavered::wave_reduce_f32;
using wavered::wave_min_i32;
using wavered::MaxOp;


constexpr int kNone = 0x7fffffff;
constexpr int kFlagged = -2;          // in out[b][0]: the item read a NaN / +inf (uniform_repair_kernel decodes it again)

// ---- the reductions OFF the dependent chain ----------------------------------------------------------------------------
// (Rounds 1-3 ran the recurrence as written above: one workgroup-wide (max, first index) reduction and one barrier per
// timestep, 52-56 % of the HBM peak at 512 x 500 x 1440.)  x -> fl(x + a) is monotone, so a maximum commutes with it:  max_i fl(post[i] + c) = fl(max_i post[i] + c)  and
// max_j fl(obs[t,j] + m) = fl(max_j obs[t,j] + m).  With  M_t = max_j obs[t,j]  (a property of the observation row alone)
// the recurrence above collapses to two scalar additions per timestep,
//     m_{t+1} = fl(fl(M_t + m_t) + c)            (t >= 1;  t = 0:  fl(max_i fl(obs[0,i] + initial[i]) + c)),
// and everything that costs a pass over a row -- M_t, and the backpointer  k_{t+1} = first i with
// fl(fl(obs[t,i] + m_t) + c) == m_{t+1}  (the same candidates, the same first-index rule, viterbi.cpp:94-100) -- is
// independent from row to row.  A workgroup (one per batch item) takes 4 R rows at a time: every wave reduces ITS rows by
// itself (DPP, no exchange), the row maxima meet in the LDS (one barrier per 4 R timesteps instead of one per timestep),
// every thread runs the 4 R-step scalar chain, then every wave finds the backpointers of its rows; the next 4 R rows are
// in flight meanwhile.  Identical indices (the candidates are the reference's, only the order of evaluation changed).
// PROBS: the observations are PROBABILITIES as from_probabilities receives them by default; every element goes through
// the reference's log() and epsilon round trip log(exp(x) + tiny) (torbi/core.py:189-197: the same three library calls, the
// same roundings as torch's elementwise kernels) on its way into the registers -- the 2 x 1.47 GB pass that otherwise
// precedes the decode is gone.  (Out of place like upstream's torch.log: the caller's tensor is not written.)
__device__ __forceinline__ float score_of_probability(float p) { return logf(expf(logf(p)) + 1.17549435e-38f); }


// An item that reads a NaN or +inf: the reference's own order of evaluation (viterbi.cpp:94-100: the running maximum of a row
// starts at prev-state 0 and is replaced on a strict '>': a NaN candidate at prev-state 0 is never replaced, a NaN anywhere
// else never wins; :218: ATen's argmax takes the FIRST NaN of the final row), with the constant matrix.  One workgroup, a
// few barriers per timestep: slow, identical to the reference operator.  Every thread of the workgroup calls it.
template <bool PROBS>
__device__ inline void faithful_uniform_item(const float *__restrict__ o, int f, const float *__restrict__ initial, float c,
                                             int32_t *__restrict__ res, int T, int S, float *red_v, int *red_k) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    auto value = [&](int t, int i) { const float x = o[(size_t)t * S + i]; return PROBS ? score_of_probability(x) : x; };
    // post_t[i] = t == 0 ? obs[0][i] + initial[i] : obs[t][i] + m
    auto post = [&](int t, int i, float m) { return t == 0 ? value(0, i) + initial[i] : value(t, i) + m; };
    float m = 0.f;
    for (int t = 1; t < f; ++t) {
        // candidates of timestep t: fl(post_{t-1}[i] + c); largest non-NaN one and its first index
        float best = -INFINITY;
        for (int i = tid; i < S; i += nt) best = __builtin_fmaxf(best, post(t - 1, i, m) + c);
        best = wave_reduce_f32(best, MaxOp());
        if (lane == 0) red_v[wave] = best;
        __syncthreads();
        best = red_v[0];
        for (int w = 1; w < nw; ++w) best = __builtin_fmaxf(best, red_v[w]);
        int k = kNone;
        for (int i = tid; i < S; i += nt)
            if (post(t - 1, i, m) + c == best) { k = i; break; }
        k = wave_min_i32(k);
        if (lane == 0) red_k[wave] = k;
        __syncthreads();
        k = red_k[0];
        for (int w = 1; w < nw; ++w) k = min(k, red_k[w]);
        const float first = post(t - 1, 0, m) + c;
        if (first != first) { best = first; k = 0; }             // the NaN at prev-state 0 stays
        if (k == kNone) k = 0;                                    // (every other candidate NaN: the first one stands)
        if (tid == 0) res[t - 1] = k;                             // the backpointer of EVERY next state (viterbi.cpp:153-157)
        m = best;
        __syncthreads();
    }
    // final state: the first NaN of the last posterior row, else its first maximum (viterbi.cpp:218-221)
    int nan_at = kNone;
    float best = -INFINITY;
    for (int i = tid; i < S; i += nt) {
        const float p = post(f - 1, i, m);
        if (p != p) nan_at = min(nan_at, i);
        best = __builtin_fmaxf(best, p);
    }
    nan_at = wave_min_i32(nan_at);
    best = wave_reduce_f32(best, MaxOp());
    if (lane == 0) { red_k[wave] = nan_at; red_v[wave] = best; }
    __syncthreads();
    nan_at = red_k[0];
    best = red_v[0];
    for (int w = 1; w < nw; ++w) { nan_at = min(nan_at, red_k[w]); best = __builtin_fmaxf(best, red_v[w]); }
    __syncthreads();
    int k = kNone;
    for (int i = tid; i < S; i += nt)
        if (post(f - 1, i, m) == best) { k = i; break; }
    k = wave_min_i32(k);
    if (lane == 0) red_k[wave] = k;
    __syncthreads();
    k = red_k[0];
    for (int w = 1; w < nw; ++w) k = min(k, red_k[w]);
    k = nan_at != kNone ? nan_at : (k == kNone ? 0 : k);
    for (int tt = f - 1 + tid; tt < T; tt += nt) res[tt] = k;
}

// NW: waves per workgroup (4; 8 or 16 for a handful of sequences, whose only parallelism is the rows in flight per item:
// 1 x 500 x 1440 0.148 ms with 4 waves)
template <int NQW, int R, bool PROBS = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void uniform_rows_kernel(const float *__restrict__ obs, const int32_t *__restrict__ frames,
                                                           const float *__restrict__ initial, float c,
                                                           int32_t *__restrict__ out, int B, int T, int S) {
    constexpr int CH = NW * R;
    __shared__ float rowmax[2][CH];
    bool odd = false;                  // a NaN / +inf among the item's inputs (nonfinite.hpp): decoded again below
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int f = frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    const float *o = obs + (size_t)b * T * S;
    int32_t *res = out + (size_t)b * T;
    const float ninf = -INFINITY;

    float4 cur[R][NQW], nxt[R][NQW];
    auto load = [&](float4 (&dst)[R][NQW], int first) {       // rows first + R wave + r, clamped to valid memory
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = first + R * wave + r;
            const float *row = o + (size_t)(t < T ? t : T - 1) * S;
#pragma unroll
            for (int q = 0; q < NQW; ++q) {
                const int i = 4 * lane + 256 * q;
                float4 v = *reinterpret_cast<const float4 *>(row + (i < S ? i : 0));
                if constexpr (PROBS)
                    v = make_float4(score_of_probability(v.x), score_of_probability(v.y), score_of_probability(v.z),
                                    score_of_probability(v.w));
                dst[r][q] = v;
            }
        }
    };
    load(cur, 0);
    if (wave == 0) {                                           // t = 0: post = obs[0] + initial (viterbi.cpp:72-76)
#pragma unroll
        for (int q = 0; q < NQW; ++q) {
            const int i = 4 * lane + 256 * q;
            const float4 a = *reinterpret_cast<const float4 *>(initial + (i < S ? i : 0));
            odd = odd || (i < S && nonfinite::odd4(a));
            cur[0][q] = make_float4(cur[0][q].x + a.x, cur[0][q].y + a.y, cur[0][q].z + a.z, cur[0][q].w + a.w);
        }
    }
    float m = 0.f;                                             // m_t of the first row of the chunk (unused at t = 0)
    int parity = 0;
    for (int t0 = 0; t0 < f; t0 += CH) {
        load(nxt, t0 + CH);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float lm = ninf;
            const bool inside = t0 + R * wave + r < f;         // (rows at or past the item's length are not its input)
#pragma unroll
            for (int q = 0; q < NQW; ++q)
                if (4 * lane + 256 * q < S) {
                    // (looked at HERE, where the row is consumed anyway: in the load it would wait for the load)
                    odd = odd || (inside && nonfinite::odd4(cur[r][q]));
                    lm = __builtin_fmaxf(__builtin_fmaxf(lm, __builtin_fmaxf(cur[r][q].x, cur[r][q].y)),
                                         __builtin_fmaxf(cur[r][q].z, cur[r][q].w));
                }
            const float wm = wave_reduce_f32(lm, MaxOp());
            if (lane == 0) rowmax[parity][R * wave + r] = wm;
        }
        __syncthreads();
        float mt[R], mt1[R], top[R];                           // m_t, m_{t+1} and max posterior of this wave's rows
        float mm = m;
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const float M = rowmax[parity][s];
            const float maxpost = t0 + s == 0 ? M : M + mm;
            const float next = maxpost + c;
            if (s / R == wave) {                               // (wave-uniform)
                mt[s % R] = mm;
                mt1[s % R] = next;
                top[s % R] = maxpost;
            }
            mm = next;
        }
        m = mm;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = t0 + R * wave + r;
            if (t <= f - 1) {                                  // (wave-uniform)
                const bool last = t == f - 1;
                const float add = t == 0 ? 0.f : mt[r];        // (row 0 already holds obs + initial; x + 0 = x)
                const float want = last ? top[r] : mt1[r];
                int lk = kNone;
#pragma unroll
                for (int q = NQW - 1; q >= 0; --q) {
                    const int i = 4 * lane + 256 * q;
                    if (i < S) {
                        float4 v = cur[r][q];
                        if (t != 0) v = make_float4(v.x + add, v.y + add, v.z + add, v.w + add);
                        if (!last) v = make_float4(v.x + c, v.y + c, v.z + c, v.w + c);
                        int kq = v.w == want ? i + 3 : kNone;
                        kq = v.z == want ? i + 2 : kq;
                        kq = v.y == want ? i + 1 : kq;
                        kq = v.x == want ? i : kq;
                        lk = min(lk, kq);
                    }
                }
                int k = wave_min_i32(lk);
                k = k == kNone ? 0 : k;
                if (!last) {
                    if (lane == 0) res[t] = k;                 // out[t] = k_{t+1}: the backpointer of every next state
                } else {
                    for (int tt = f - 1 + lane; tt < T; tt += 64) res[tt] = k;     // (viterbi.cpp:218-221)
                }
            }
        }
        parity ^= 1;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < NQW; ++q) cur[r][q] = nxt[r][q];
    }
    // an item that read a NaN / +inf says so in its first output position; uniform_repair_kernel, queued behind this launch,
    // decodes it again (in here the reference-order path costs the streaming kernel a third of its occupancy)
    odd = odd || nonfinite::odd(c);
    if (__syncthreads_or(odd) && tid == 0) res[0] = kFlagged;
}

// grid = B, block = 256: items uniform_rows_kernel flagged, again in the reference's order of evaluation
template <bool PROBS>
__global__ __launch_bounds__(256) void uniform_repair_kernel(const float *__restrict__ obs, const int32_t *__restrict__ frames,
                                                             const float *__restrict__ initial, float c,
                                                             int32_t *__restrict__ out, int B, int T, int S) {
    __shared__ float red_v[4];
    __shared__ int red_k[4];
    const int b = blockIdx.x;
    int32_t *res = out + (size_t)b * T;
    if (res[0] != kFlagged) return;
    int f = frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    __syncthreads();                         // (every thread has read the flag before thread 0 overwrites it)
    faithful_uniform_item<PROBS>(obs + (size_t)b * T * S, f, initial, c, res, T, S, red_v, red_k);
}

}  // namespace uniform
