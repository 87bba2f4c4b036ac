// nonfinite.hpp -- NaN and +inf inputs: the reference's results, whatever route the decode took.
//
// The reference CPU operator is deterministic on NaN (torbi/csrc/viterbi.cpp:94-100: the running maximum of a row starts at
// prev-state 0 and is replaced on a strict '>', so a NaN candidate at prev-state 0 is never replaced and a NaN candidate
// anywhere else never wins; :218: ATen's argmax takes the FIRST NaN of the final row), and +inf meets -inf as NaN.  The fast
// routes evaluate maxima with v_max_f32 (a NaN operand loses wherever it stands), prune with bounds, recompute argmaxima
// lazily: on such inputs they would silently decode something else.  So:
//   * every forward kernel looks at the posterior values it produces anyway -- post'[j] = fl(obs[t][j] + max) is NaN or
//     +inf whenever the observation is (one compare per VALUE, against hundreds of cells per value) -- and raises the batch's
//     alarm word; the matrix and the initial vector are looked at by a small launch of their own (8.3 MB at 1440 states);
//     the routes that launch a kernel per timestep look at the observations in that launch too;
//   * a launch behind the decode does nothing unless an alarm was raised; then every item of the batch looks for NaN / +inf
//     in ITS inputs (observation rows below its length, matrix, initial), and an item that has one is decoded AGAIN by one
//     workgroup exactly as viterbi.cpp:65-108, 140-160, 218-221 does it: scan order, strict '>', int32 trellis, ATen's
//     argmax -- slow (a second or so per thousand timesteps at 1440 states) and identical to the reference operator.
// An alarm is "raised" by storing the decode's serial number (host counter): no word has to be cleared between decodes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace nonfinite {

constexpr int kAlarmWord = 4;            // of the route record: observations / posterior values
constexpr int kMatrixWord = 5;           // ... transition matrix / initial vector (every item again)
constexpr int kBandWord = 6;             // ... a band with a constant outside whose tests did not decide (band_tile_forward.hpp):
                                         //     every item of the batch again
constexpr int kMaxBatches = 16;

__device__ __forceinline__ bool odd(float x) { return !(x <= 3.402823466e+38f); }        // NaN or +inf (-inf is in contract)
__device__ __forceinline__ bool odd4(const float4 &v) { return odd(v.x) || odd(v.y) || odd(v.z) || odd(v.w); }
// (wave-uniform store: called where every lane of the wave arrives)
__device__ __forceinline__ void raise(bool seen, int32_t *alarm, int serial) {
    if (__any(seen) && (threadIdx.x & 63) == 0 && alarm) *alarm = serial;
}

struct Records {
    int32_t *record[kMaxBatches];        // route records of the launch group's batches
    int n;
};

// matrix and initial vector: grid = up to 1024 blocks of 256.  `reach_left` >= 0: the caller of the band route promised that
// trans[j][i] is -inf unless j - reach_left <= i <= j + reach_right (include/torbi_hip.h, torbi_hip_viterbi_decode_banded);
// the band kernels never read outside it, so an entry there that is NOT -inf -- a stale promise: the matrix was edited
// behind a cached look -- raises the same alarm, and every item is decoded again on the whole matrix.  `background`: what the
// promise says every entry outside the band is (-inf, or the one constant of band_tile_forward.hpp), compared bit for bit.
__global__ __launch_bounds__(256) void matrix_kernel(const float *__restrict__ trans, const float *__restrict__ initial, int S,
                                                     Records recs, int serial, int reach_left, int reach_right, float background) {
    const size_t n = (size_t)S * S;
    bool seen = false;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float x = trans[e];
        seen = seen || odd(x);
        if (reach_left >= 0) {
            const int j = (int)(e / (size_t)S), i = (int)(e - (size_t)j * S);
            seen = seen || ((i < j - reach_left || i > j + reach_right) && __float_as_uint(x) != __float_as_uint(background));
        }
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < S; i += 256) seen = seen || odd(initial[i]);
    if (__any(seen) && (threadIdx.x & 63) == 0)
        for (int k = 0; k < recs.n; ++k) recs.record[k][kMatrixWord] = serial;
}

// observations of one batch (the routes that do not look themselves): grid = up to 2048 blocks of 256
__global__ __launch_bounds__(256) void observation_kernel(const float *__restrict__ obs, size_t count, int32_t *__restrict__ record,
                                                          int serial) {
    bool seen = false;
    if ((reinterpret_cast<uintptr_t>(obs) & 15) == 0) {
        const float4 *p = reinterpret_cast<const float4 *>(obs);
        const size_t n4 = count / 4;
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) seen = seen || odd4(p[e]);
        for (size_t e = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (size_t)gridDim.x * 256)
            seen = seen || odd(obs[e]);
    } else {
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (size_t)gridDim.x * 256) seen = seen || odd(obs[e]);
    }
    raise(seen, record + kAlarmWord, serial);
}

// One item, one workgroup, the reference's arithmetic in the reference's order (viterbi.cpp:65-108, 140-160, 218-221).
// obs: the item's [T][S]; trellis: [T][S] int32; rows: [2][S] scratch; out: [T].  Every thread of the workgroup calls it.
__device__ inline void faithful_item(const float *__restrict__ obs, int f, const float *__restrict__ trans,
                                     const float *__restrict__ initial, int32_t *__restrict__ trellis, float *__restrict__ rows,
                                     int32_t *__restrict__ out, int T, int S) {
    const int tid = threadIdx.x, nt = blockDim.x;
    float *cur = rows, *nxt = rows + S;
    for (int i = tid; i < S; i += nt) cur[i] = obs[i] + initial[i];                 // viterbi.cpp:72-76
    __syncthreads();
    for (int t = 1; t < f; ++t) {
        for (int j = tid; j < S; j += nt) {
            const float *tr = trans + (size_t)j * S;
            float best = cur[0] + tr[0];                                           // :94: the running maximum starts at prev-state 0
            int arg = 0;
            for (int i = 1; i < S; ++i) {
                const float c = cur[i] + tr[i];                                    // :84
                if (c > best) { best = c; arg = i; }                               // :97-100 (a NaN never passes, a NaN `best` never yields)
            }
            trellis[(size_t)t * S + j] = arg;
            nxt[j] = obs[(size_t)t * S + j] + best;                                // :102
        }
        __syncthreads();
        float *swap = cur; cur = nxt; nxt = swap;
    }
    if (tid == 0) {
        int arg = 0;                                                               // :218, ATen: the first NaN, else the first maximum
        float best = cur[0];
        for (int i = 1; i < S; ++i)
            if (cur[i] > best || (cur[i] != cur[i] && best == best)) { best = cur[i]; arg = i; }
        for (int t = f - 1; t < T; ++t) out[t] = arg;                              // :219-221
        int index = arg;
        for (int t = f - 1; t >= 1; --t) {                                         // :153-157
            index = trellis[(size_t)t * S + index];
            out[t - 1] = index;
        }
    }
    __syncthreads();
    // (the value-only routes keep the final posterior row in the history: torbi_hip_read_posterior reads it there)
    float *last = reinterpret_cast<float *>(trellis) + (size_t)(f - 1) * S;
    for (int i = tid; i < S; i += nt) last[i] = cur[i];
}

struct RepairJobs {
    const float *obs[kMaxBatches];
    const int32_t *frames[kMaxBatches];
    int32_t *out[kMaxBatches];
    int32_t *trellis[kMaxBatches];       // offset 0 of the batch's workspace: [B][T][S] on every route
    float *rows[kMaxBatches];            // [B][2][S]
    int32_t *record[kMaxBatches];
    int B[kMaxBatches], T[kMaxBatches], item0[kMaxBatches];
    int n;
};

// grid = items of the launch group, block = 256
__global__ __launch_bounds__(256) void repair_kernel(RepairJobs jobs, const float *__restrict__ trans, const float *__restrict__ initial,
                                                     int S, int serial) {
    int k = 0;
    while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.item0[k + 1]) ++k;
    const int32_t *record = jobs.record[k];
    const bool matrix = record[kMatrixWord] == serial || record[kBandWord] == serial;
    if (!matrix && record[kAlarmWord] != serial) return;
    const int b = (int)blockIdx.x - jobs.item0[k], T = jobs.T[k];
    int f = jobs.frames[k][b];
    f = f < 1 ? 1 : (f > T ? T : f);
    const float *obs = jobs.obs[k] + (size_t)b * T * S;
    bool seen = matrix;
    if (!seen) {                        // does THIS item read a NaN / +inf?  (rows below its length only: viterbi.cpp:67-78)
        const size_t n = (size_t)f * S;
        for (size_t e = threadIdx.x; e < n; e += 256) seen = seen || odd(obs[e]);
    }
    if (!__syncthreads_or(seen)) return;
    faithful_item(obs, f, trans, initial, jobs.trellis[k] + (size_t)b * T * S, jobs.rows[k] + (size_t)b * 2 * S,
                  jobs.out[k] + (size_t)b * T, T, S);
}

}  // namespace nonfinite
