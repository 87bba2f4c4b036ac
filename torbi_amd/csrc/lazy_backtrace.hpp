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
__global__ __launch_bounds__(64) void backtrace_kernel(const float *__restrict__ hist,
                                                       const float *__restrict__ trans,
                                                       const int32_t *__restrict__ frames,
                                                       int32_t *__restrict__ out, int B, int T, int S) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    int f = frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    const float *h = hist + (size_t)b * T * S;
    int32_t *o = out + (size_t)b * T;

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

// Register-resident form for S % 4 == 0 and S <= 256*NQ: the posterior row of the NEXT path step
// does not depend on the state being resolved, so it is prefetched into registers while the current
// step's transition row (which does) is in flight; one wave per batch item, lanes own 4 consecutive
// prev-states per 256-wide stripe (ascending per lane, as the reference scan).
template <int NQ>
__global__ __launch_bounds__(64) void backtrace_prefetch_kernel(const float *__restrict__ hist,
                                                                const float *__restrict__ trans,
                                                                const int32_t *__restrict__ frames,
                                                                int32_t *__restrict__ out, int B, int T, int S) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    int f = frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    const float *h = hist + (size_t)b * T * S;
    int32_t *o = out + (size_t)b * T;
    const float4 ninf = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);

    float4 cur[NQ], nxt[NQ];
    {
        const float *row = h + (size_t)(f - 1) * S;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            cur[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : ninf;
        }
    }
    if (f >= 2) {
        const float *row = h + (size_t)(f - 2) * S;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            nxt[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : ninf;
        }
    }
    float best = -INFINITY;
    int arg = kSentinel;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = 4 * lane + 256 * q;
        if (i < S) {
            scan1(cur[q].x, i, best, arg); scan1(cur[q].y, i + 1, best, arg);
            scan1(cur[q].z, i + 2, best, arg); scan1(cur[q].w, i + 3, best, arg);
        }
    }
    int j = wave_first_argmax(best, arg);
    for (int tt = f - 1 + lane; tt < T; tt += 64) o[tt] = j;

    for (int tt = f - 1; tt >= 1; --tt) {
        // transition row of the state just resolved (depends on j) ...
        const float *tr = trans + (size_t)j * S;
        float4 q4[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            q4[q] = i < S ? *reinterpret_cast<const float4 *>(tr + i) : ninf;
        }
        // ... posterior row tt-1 is already in registers; fetch row tt-2 for the next step
#pragma unroll
        for (int q = 0; q < NQ; ++q) cur[q] = nxt[q];
        if (tt >= 2) {
            const float *row = h + (size_t)(tt - 2) * S;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int i = 4 * lane + 256 * q;
                nxt[q] = i < S ? *reinterpret_cast<const float4 *>(row + i) : ninf;
            }
        }
        best = -INFINITY;
        arg = kSentinel;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = 4 * lane + 256 * q;
            if (i < S) {
                scan1(cur[q].x + q4[q].x, i, best, arg); scan1(cur[q].y + q4[q].y, i + 1, best, arg);
                scan1(cur[q].z + q4[q].z, i + 2, best, arg); scan1(cur[q].w + q4[q].w, i + 3, best, arg);
            }
        }
        j = wave_first_argmax(best, arg);
        if (lane == 0) o[tt - 1] = j;
    }
}

}  // namespace lazy
