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
// observations once and is HBM-bound (4S + 4 bytes per timestep).  One wave per batch item; the
// posterior row lives in registers; observation rows are prefetched DEPTH timesteps ahead.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "wave_reduce.hpp"

namespace uniform {

using wavered::wave_reduce_f32;
using wavered::wave_min_i32;
using wavered::MaxOp;


constexpr int kNone = 0x7fffffff;

// first index (ascending) whose value equals m among this lane's 4*NQ elements, else kNone
template <int NQ>
__device__ __forceinline__ int first_equal(const float4 (&v)[NQ], float m, int lane, int S) {
    int k = kNone;
#pragma unroll
    for (int q = NQ - 1; q >= 0; --q) {
        const int i = 4 * lane + 256 * q;
        if (i < S) {
            k = v[q].w == m ? i + 3 : k;
            k = v[q].z == m ? i + 2 : k;
            k = v[q].y == m ? i + 1 : k;
            k = v[q].x == m ? i : k;
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

template <int NQ>
__device__ __forceinline__ void load_row(float4 (&r)[NQ], const float *row, int lane, int S) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = 4 * lane + 256 * q;
        r[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// One workgroup of 4 waves per batch item; thread `tid` owns prev-states {4*tid + 1024*q}, q < NQ
// (S <= 1024*NQ, S % 4 == 0); DEPTH observation rows in flight.  Per timestep each wave reduces its
// own (max, first index of that max) with DPP, the four pairs meet in LDS (one barrier, slots
// double-buffered by timestep parity) and every wave forms
//     m = max_w m_w,   k = min { k_w : m_w == m }        (lowest index among equal maxima).
template <int NQ, int DEPTH>
__global__ __launch_bounds__(256) void uniform_decode_kernel(const float *__restrict__ obs,
                                                             const int32_t *__restrict__ frames,
                                                             const float *__restrict__ initial, float c,
                                                             int32_t *__restrict__ out, int B, int T, int S) {
    __shared__ float xm[2][4];
    __shared__ int xk[2][4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int f = frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    const float *o = obs + (size_t)b * T * S;
    int32_t *res = out + (size_t)b * T;

    float4 post[NQ], cand[NQ], rows[DEPTH][NQ];
    // element addressing shared by every helper below: q-th float4 of this thread
#define U_IDX(q) (4 * tid + 1024 * (q))
#define U_LOAD(dst, row)                                                                     \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                          \
        (dst)[q] = U_IDX(q) < S ? *reinterpret_cast<const float4 *>((row) + U_IDX(q))      \
                                : make_float4(0.f, 0.f, 0.f, 0.f)
    // block-wide (max, first index attaining it) of v[]: returns m, writes k
    int parity = 0;   // exchange slots alternate on EVERY reduction (a slot is rewritten only after
                      // the barrier of the next reduction, i.e. after every wave has read it)
    auto reduce = [&](const float4 (&v)[NQ], int &k_out) -> float {
        float lm = -INFINITY;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (U_IDX(q) < S)
                lm = __builtin_fmaxf(__builtin_fmaxf(lm, __builtin_fmaxf(v[q].x, v[q].y)),
                                     __builtin_fmaxf(v[q].z, v[q].w));
        const float wm = wave_reduce_f32(lm, MaxOp());
        int lk = kNone;
#pragma unroll
        for (int q = NQ - 1; q >= 0; --q) {
            const int i = U_IDX(q);
            if (i < S) {
                int kq = v[q].w == wm ? i + 3 : kNone;
                kq = v[q].z == wm ? i + 2 : kq;
                kq = v[q].y == wm ? i + 1 : kq;
                kq = v[q].x == wm ? i : kq;
                lk = min(lk, kq);
            }
        }
        const int wk = wave_min_i32(lk);
        if (lane == 0) { xm[parity][wave] = wm; xk[parity][wave] = wk; }
        __syncthreads();
        const float m0 = xm[parity][0], m1 = xm[parity][1], m2 = xm[parity][2], m3 = xm[parity][3];
        const float m = __builtin_fmaxf(__builtin_fmaxf(m0, m1), __builtin_fmaxf(m2, m3));
        int k = m0 == m ? xk[parity][0] : kNone;
        k = min(k, m1 == m ? xk[parity][1] : kNone);
        k = min(k, m2 == m ? xk[parity][2] : kNone);
        k = min(k, m3 == m ? xk[parity][3] : kNone);
        k_out = k;
        parity ^= 1;
        return m;
    };

    // t = 0: post = obs[0] + initial                                        (viterbi.cpp:72-76)
    {
        float4 a[NQ], i4[NQ];
        U_LOAD(a, o);
        U_LOAD(i4, initial);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            post[q] = make_float4(a[q].x + i4[q].x, a[q].y + i4[q].y, a[q].z + i4[q].z, a[q].w + i4[q].w);
    }
    // Observation rows are always fetched (row index clamped to T-1, valid memory), so the
    // unrolled body has no conditional register-array writes (those would be demoted to scratch).
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const int r = 1 + d < T ? 1 + d : T - 1;
        U_LOAD(rows[d], o + (size_t)r * S);
    }

    int t0 = 1;
    for (; t0 + DEPTH <= f; t0 += DEPTH) {                    // whole groups: no per-step test
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = t0 + d;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                cand[q] = make_float4(post[q].x + c, post[q].y + c, post[q].z + c, post[q].w + c);
            int k;
            const float m = reduce(cand, k);
            if (tid == 0) res[t - 1] = k;                     // the backpointer of every next state
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                post[q] = make_float4(rows[d][q].x + m, rows[d][q].y + m, rows[d][q].z + m, rows[d][q].w + m);
            const int r = t + DEPTH < T ? t + DEPTH : T - 1;
            U_LOAD(rows[d], o + (size_t)r * S);
        }
    }
    // remainder (< DEPTH steps): same body, the new posterior is committed with a select
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) {
        const int t = t0 + d;
        const bool live = t < f;                              // block-uniform
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            cand[q] = make_float4(post[q].x + c, post[q].y + c, post[q].z + c, post[q].w + c);
        int k;
        const float m = reduce(cand, k);
        if (live && tid == 0) res[t - 1] = k;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            post[q].x = live ? rows[d][q].x + m : post[q].x;
            post[q].y = live ? rows[d][q].y + m : post[q].y;
            post[q].z = live ? rows[d][q].z + m : post[q].z;
            post[q].w = live ? rows[d][q].w + m : post[q].w;
        }
    }
    // final state = first argmax of the last posterior row (viterbi.cpp:218); it fills every
    // position t >= frames-1 (viterbi.cpp:219-221)
    int fin;
    (void)reduce(post, fin);
    for (int tt = f - 1 + tid; tt < T; tt += 256) res[tt] = fin;
#undef U_IDX
#undef U_LOAD
}

}  // namespace uniform
