// Up to 256 states: forward recurrence AND backtrace of a sequence in ONE launch, the transition matrix in registers --
// one wavefront per sequence up to 64 states (first half of this file), one workgroup up to 256 (second half).
//
// The reference runs these shapes through the same trellis kernels as any other (viterbi.cu:203-241: one block per item,
// one `__syncthreads` round per timestep); per-timestep LAUNCHES cost 4-5 us each here whatever the state count
// (torbi_hip.hip, launch_forward), so a 3-state toy or a 40-class posteriorgram paid for launches only.  With S <= 64 a
// wavefront holds everything: lane j owns next-state j -- its transition row trans[j][*] in registers, its posterior in one
// VGPR -- and a timestep is the previous row broadcast through the LDS, then S x (add -> strict '>' compare -> select ->
// max) with the reference's tie rule (viterbi.cpp:91-104, viterbi.cu:82-123: lowest prev-state wins, backpointer 0 unless
// replaced).  Backpointers are bytes, four timesteps to a dword per lane, in a plane at the start of the workspace; the
// same wavefront then walks them back from the first maximum of the last posterior row (viterbi.cpp:153-157, :218-221):
// the rows of 64 timesteps are loaded at once (they do not depend on the path), a path step is one v_readlane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#include "wave_reduce.hpp"
#include "nonfinite.hpp"

namespace small {

constexpr int kMaxS = 64;
// (4, or the next multiple of 8: never more than 2 S, so the backpointer plane fits where the int32 trellis would lie)
__host__ __device__ inline int padded_states(int S) { return S <= 4 ? 4 : (S + 7) / 8 * 8; }
// dwords of the backpointer plane per item: [ceil((T-1)/4)][SP]
__host__ __device__ inline size_t plane_dwords(int T, int S) { return (size_t)((T - 1 + 3) / 4) * padded_states(S); }
inline bool supported(int S) { return S >= 2 && S <= kMaxS; }

// SP: padded state count (transition entries beyond S are -inf: never a maximum); CH: timesteps whose
// observation rows are in flight while the previous CH are computed (a multiple of 4)
template <int SP, int CH>
__global__ __launch_bounds__(64) void decode_kernel(const float *__restrict__ obs, const int32_t *__restrict__ frames,
                                                    const float *__restrict__ trans, const float *__restrict__ init,
                                                    int32_t *__restrict__ out, uint32_t *__restrict__ plane,
                                                    float *__restrict__ post0, float *__restrict__ post1,
                                                    int32_t *__restrict__ route_record, int route, int B, int T, int S, int serial) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (b == 0 && lane == 0) *route_record = route;
    int n = __builtin_amdgcn_readfirstlane(frames[b]);
    n = n < 1 ? 1 : (n > T ? T : n);
    const bool live = lane < S;
    const float ninf = -__builtin_huge_valf();

    float row[SP];                                   // trans[lane][i]
#pragma unroll
    for (int i = 0; i < SP; ++i) row[i] = trans[(size_t)min(lane, S - 1) * S + min(i, S - 1)];   // (clamped addresses:
    asm volatile("" ::: "memory");                   //  every load unconditional and in flight before the first use)
    bool odd_matrix = false;                         // the matrix / initial vector hold a NaN or +inf (nonfinite.hpp: this
#pragma unroll                                       // route looks itself, its launches are too short for a launch that does)
    for (int i = 0; i < SP; ++i) odd_matrix = odd_matrix || nonfinite::odd(row[i]);
    odd_matrix = odd_matrix || nonfinite::odd(init[min(lane, S - 1)]);
    nonfinite::raise(odd_matrix, route_record + nonfinite::kMatrixWord, serial);
#pragma unroll
    for (int i = 0; i < SP; ++i) row[i] = fminf(row[i], (live && i < S) ? -ninf : ninf);             // the padding: -inf
    const float *o = obs + (size_t)b * T * S + min(lane, S - 1);
    uint32_t *pl = plane + (size_t)b * plane_dwords(T, S);
    float p = o[0] + init[min(lane, S - 1)];
    p = live ? p : ninf;
    bool odd = nonfinite::odd(p);                    // a NaN / +inf posterior value was produced (nonfinite.hpp)

    float cur[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) cur[k] = o[(size_t)min(1 + k, n - 1) * S];
    for (int t0 = 1; t0 < n; t0 += CH) {
        float nxt[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) nxt[k] = o[(size_t)min(t0 + CH + k, n - 1) * S];       // (clamped: never past the item)
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
            if (t0 + 4 * q >= n) break;
            uint32_t packed = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * q + r;
                if (t0 + k < n) {                    // (wave-uniform)
                    // the row through the LDS: one broadcast ds_read_b128 per four prev-states (v_readlane into a scalar
                    // register costs an instruction and a wait state per prev-state: 4.7 against 3.0 ms at 1 x 5000 x 64).
                    // NC running maxima over consecutive ranges of prev-states (a lone wavefront is bound by the dependent
                    // compare / select chain otherwise), merged in index order with the same strict '>': the lowest prev-state
                    // still wins every tie.  (Left to the compiler's scheduling on purpose: the fixed four_cells sequence the
                    // workgroup kernel below needs is 15-40 % slower here, where nothing else hides a latency.)
                    constexpr int NC = SP >= 32 ? 4 : SP >= 16 ? 2 : 1;
                    constexpr int L = SP / NC;
                    __shared__ float4 shared_row[SP / 4];
                    if (lane < SP) reinterpret_cast<float *>(shared_row)[lane] = p;
                    __builtin_amdgcn_wave_barrier();
                    float pv[SP];
#pragma unroll
                    for (int i = 0; i < SP / 4; ++i) {
                        const float4 v = shared_row[i];
                        pv[4 * i] = v.x; pv[4 * i + 1] = v.y; pv[4 * i + 2] = v.z; pv[4 * i + 3] = v.w;
                    }
                    __builtin_amdgcn_wave_barrier();
                    float bestc[NC];
                    uint32_t argc[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        bestc[c] = pv[c * L] + row[c * L];
                        argc[c] = (uint32_t)(c * L);
                    }
#pragma unroll
                    for (int e = 1; e < L; ++e) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const int i = c * L + e;
                            const float cand = pv[i] + row[i];
                            argc[c] = cand > bestc[c] ? (uint32_t)i : argc[c];
                            bestc[c] = fmaxf(bestc[c], cand);
                        }
                    }
                    float best = bestc[0];
                    uint32_t arg = argc[0];
#pragma unroll
                    for (int c = 1; c < NC; ++c) {
                        const bool better = bestc[c] > best;
                        best = better ? bestc[c] : best;
                        arg = better ? argc[c] : arg;
                    }
                    p = live ? cur[k] + best : ninf;
                    odd = odd || nonfinite::odd(p);
                    packed |= arg << (8 * r);
                }
            }
            if (lane < SP) pl[(size_t)((t0 - 1) / 4 + q) * SP + lane] = packed;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) cur[k] = nxt[k];
    }
    if (live) (((n - 1) & 1) ? post1 : post0)[(size_t)b * S + lane] = p;      // (where torbi_hip_read_posterior looks)
    nonfinite::raise(odd, route_record + nonfinite::kAlarmWord, serial);

    // first maximum of the last row
    float best = p;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = fmaxf(best, __shfl_xor(best, d, 64));
    const unsigned long long at = __ballot(live && p == best);
    int idx = at ? __ffsll((long long)at) - 1 : 0;
    idx = __builtin_amdgcn_readfirstlane(idx);
    int32_t *ob = out + (size_t)b * T;
    for (int t = n - 1 + lane; t < T; t += 64) ob[t] = idx;
    if (n < 2) return;
    __threadfence_block();                           // this wavefront's own plane stores, read back below

    // walk back: positions n-2 .. 0; group g holds the backpointers of timesteps 4g+1 .. 4g+4
    const int groups = (n - 1 + 3) / 4;
    for (int c = (groups - 1) / 16; c >= 0; --c) {
        uint32_t rows[16];
#pragma unroll
        for (int g = 0; g < 16; ++g)
            rows[g] = (16 * c + g < groups && lane < SP) ? pl[(size_t)(16 * c + g) * SP + lane] : 0u;
        int32_t mine = 0;
#pragma unroll
        for (int g = 15; g >= 0; --g) {
#pragma unroll
            for (int r = 3; r >= 0; --r) {
                const int t = 4 * (16 * c + g) + 1 + r;
                if (t <= n - 1) {                    // (wave-uniform)
                    const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)rows[g], idx);
                    idx = (int)((word >> (8 * r)) & 255u);
                    mine = lane == 4 * g + r ? idx : mine;
                }
            }
        }
        const int pos = 64 * c + lane;               // position t-1 of timestep t = 64c + lane + 1
        if (pos <= n - 2) ob[pos] = mine;
    }
}

// ---- up to 64 states, MANY sequences: the same wavefront-per-sequence decode, value-only -------------------------------
// With several wavefronts per SIMD the kernel above is bound by instruction issue: 4.3 vector instructions per cell (add,
// compare, max, select).  Here a cell is add, add, 1/2 max3 = 1.5: the forward pass keeps no backpointers, it stores the
// posterior rows (fp32: hist[b][t][:], where the int32 trellis of the generic route would lie) and the walk back recomputes
// the first argmax of fl(hist[t-1][i] + trans[j][i]) for the state j on the path alone (viterbi.cpp:81-100: strict '>'
// from prev-state 0 = the lowest index among the maxima; every candidate -inf = the zero default).  The walk needs row j of
// the matrix per step: the four sequences of a workgroup share one copy of it in the LDS (16 KB at 64 states), lane i reads
// trans[j][i] -- one conflict-free ds_read_b32 behind the v_readfirstlane of the step before.  A lone wavefront gains
// nothing from this (its walk back is a dependent chain of ~300 cycles per step against ~10 through byte backpointers):
// the launcher takes this kernel when the batch keeps every SIMD busy with several sequences (torbi_hip.hip, launch_small).
#ifndef SMALL_ABL
#define SMALL_ABL 0          // timing only (tools/small_abl_probe.py): 1 no walk back, 2 no history stores, 4 no LDS broadcast,
#endif                       // 8 no observation loads
template <int SP, int CH>
__global__ __launch_bounds__(256) void decode_value_kernel(const float *__restrict__ obs, const int32_t *__restrict__ frames,
                                                           const float *__restrict__ trans, const float *__restrict__ init,
                                                           int32_t *__restrict__ out, float *__restrict__ hist,
                                                           float *__restrict__ post0, float *__restrict__ post1,
                                                           int32_t *__restrict__ route_record, int route, int B, int T, int S, int serial) {
    __shared__ float matrix[kMaxS * kMaxS];           // trans[j][i] at j * SP + i (the padding is never read)
    __shared__ float4 shared_rows[4][SP / 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x == 0 && tid == 0) *route_record = route;
    bool odd_matrix = false;                          // (nonfinite.hpp: this route looks at the matrix itself)
    for (int e = tid; e < S * S; e += 256) {
        const float x = trans[e];
        odd_matrix = odd_matrix || nonfinite::odd(x);
        matrix[(e / S) * SP + e % S] = x;
    }
    for (int i = tid; i < S; i += 256) odd_matrix = odd_matrix || nonfinite::odd(init[i]);
    nonfinite::raise(odd_matrix, route_record + nonfinite::kMatrixWord, serial);
    __syncthreads();
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;                               // (behind the only workgroup barrier)
    int n = __builtin_amdgcn_readfirstlane(frames[b]);
    n = n < 1 ? 1 : (n > T ? T : n);
    const bool live = lane < S;
    const float ninf = -__builtin_huge_valf();

    float row[SP];                                    // trans[lane][i]
#pragma unroll
    for (int i = 0; i < SP; ++i) row[i] = (live && i < S) ? matrix[lane * SP + i] : ninf;
    const float *o = obs + (size_t)b * T * S + min(lane, S - 1);
    float *h = hist + (size_t)b * T * S;
    float p = o[0] + init[min(lane, S - 1)];
    p = live ? p : ninf;
    bool odd = nonfinite::odd(p);                    // a NaN / +inf posterior value was produced (nonfinite.hpp)
    if (live) h[lane] = p;
    float4 *const shared_row = shared_rows[wave];

    float cur[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) cur[k] = o[(size_t)min(1 + k, n - 1) * S];
    for (int t0 = 1; t0 < n; t0 += CH) {
        float nxt[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) nxt[k] = (SMALL_ABL & 8) ? 0.25f * k : o[(size_t)min(t0 + CH + k, n - 1) * S];       // (clamped: never past the item)
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (t0 + k < n) {                         // (wave-uniform)
                if (lane < SP) reinterpret_cast<float *>(shared_row)[lane] = p;
                __builtin_amdgcn_wave_barrier();
                float pv[SP];
#pragma unroll
                for (int i = 0; i < SP / 4; ++i) {
                    float4 v;
                    if (SMALL_ABL & 4) v = make_float4(p, p + 1.f, p + 2.f, p + 3.f);
                    else v = shared_row[i];
                    pv[4 * i] = v.x; pv[4 * i + 1] = v.y; pv[4 * i + 2] = v.z; pv[4 * i + 3] = v.w;
                }
                __builtin_amdgcn_wave_barrier();
                // four running maxima over interleaved prev-states (max is exact and order independent: any tree gives the value)
                float best[4] = {ninf, ninf, ninf, ninf};
#pragma unroll
                for (int i = 0; i + 8 <= SP; i += 8) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        best[c] = fmaxf(fmaxf(best[c], pv[i + c] + row[i + c]), pv[i + 4 + c] + row[i + 4 + c]);
                }
                if (SP % 8) {                          // (SP = 4: one group of four)
#pragma unroll
                    for (int c = 0; c < 4; ++c) best[c] = fmaxf(best[c], pv[SP - 4 + c] + row[SP - 4 + c]);
                }
                const float top = fmaxf(fmaxf(best[0], best[1]), fmaxf(best[2], best[3]));
                p = live ? cur[k] + top : ninf;
                odd = odd || nonfinite::odd(p);
                if (live && !(SMALL_ABL & 2)) h[(size_t)(t0 + k) * S + lane] = p;
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) cur[k] = nxt[k];
    }
    if (live) (((n - 1) & 1) ? post1 : post0)[(size_t)b * S + lane] = p;      // (where torbi_hip_read_posterior looks)
    nonfinite::raise(odd, route_record + nonfinite::kAlarmWord, serial);
    if (SMALL_ABL & 1) return;

    // first maximum of the last row
    float best = p;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = fmaxf(best, __shfl_xor(best, d, 64));
    unsigned long long at = __ballot(live && p == best);
    int idx = at ? __ffsll((long long)at) - 1 : 0;
    idx = __builtin_amdgcn_readfirstlane(idx);
    int32_t *ob = out + (size_t)b * T;
    for (int t = n - 1 + lane; t < T; t += 64) ob[t] = idx;
    if (n < 2) return;
    __threadfence_block();                            // this wavefront's own history stores, read back below

    // walk back: timestep t -> position t - 1 holds the first argmax over i of hist[t-1][i] + trans[idx][i]; the rows of
    // 16 timesteps are asked for together (they do not depend on the path)
    for (int hi = n - 1; hi >= 1; hi -= 16) {
        float rows[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) rows[g] = (hi - g >= 1 && live) ? h[(size_t)(hi - g - 1) * S + lane] : ninf;
        int32_t mine = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (hi - g >= 1) {                        // (wave-uniform)
                const float cand = live ? rows[g] + matrix[idx * SP + lane] : ninf;
                const float m = wavered::wave_reduce_f32(cand, wavered::MaxOp());      // (DPP: no trip through the LDS crossbar)
                at = __ballot(live && cand == m);
                idx = __builtin_amdgcn_readfirstlane(at ? __ffsll((long long)at) - 1 : 0);
                mine = lane == g ? idx : mine;
            }
        }
        const int pos = hi - 1 - lane;                // position of timestep hi - lane
        if (lane < 16 && pos >= 0) ob[pos] = mine;
    }
}

// ---- 65 .. 256 states: one WORKGROUP per sequence (or per two), value-only ---------------------------------------------
// The matrix still fits the registers of one compute unit (256 x 256 x 4 B = half of its vector register file): wave
// (nb, pq) of NB x PQ owns next-states [64 nb, 64 nb + 64) x prev-states [pq L, pq L + L), L <= 64 -- lane j keeps that
// piece of row j in L registers for the whole launch, and with ~100-115 registers a lane the 4 .. 16 waves of a workgroup
// hide each other's latencies.  A timestep: the previous posterior row is broadcast from the LDS (one ds_read_b128 per four
// prev-states), four interleaved running maxima per lane keep the dependent chain short, the PQ pieces of a row meet
// through the LDS, the waves with pq = 0 add the observation and write the new row.
// (Rounds 4-5 carried a byte-backpointer form beside it -- add / compare / max / select with (value, index) merges, 4.3
// instructions per cell, the walk back in the same launch: 2.17 against 1.02 ms at 512 x 500 x 256, 1.05 against 0.49 for ONE
// sequence; removed.)
constexpr int kBlockMaxS = 256;
__host__ __device__ inline int block_splits(int S) { return (S + 63) / 64; }                          // PQ (= NB)
__host__ __device__ inline int block_row_registers(int S) {                                           // L
    return (S + block_splits(S) - 1) / block_splits(S) <= 48 ? 48 : 64;
}
inline bool block_supported(int S) { return S > kMaxS && S <= kBlockMaxS; }

// A cell is add, add, 1/2 max3: the forward pass keeps no backpointers, it stores the posterior rows (fp32: hist[b][t][:],
// where the generic route's trellis would lie) and the backtrace is a launch of its own -- lazy_backtrace.hpp recomputes
// the first argmax of fl(hist[t-1][i] + trans[j][i]) for the state on the path (viterbi.cpp:81-100), in speculative
// segments for a batch of few paths.  The pieces of a row meet through the LDS as VALUES (max is exact and order
// independent).
// NSEQ = 2: a workgroup decodes TWO sequences against the one copy of the matrix in its registers -- the two barriers and
// the merge of a timestep, which nothing overlaps when a 16-wave workgroup has a compute unit to itself, are shared by two
// independent recurrences; taken when the compute units are full without it (torbi_hip.hip, launch_block_value_as).
template <int PQ, int L, int NSEQ>
__global__ __launch_bounds__(64 * PQ * PQ) void block_value_kernel(
    const float *__restrict__ obs, const int32_t *__restrict__ frames, const float *__restrict__ trans,
    const float *__restrict__ init, float *__restrict__ hist, float *__restrict__ post0, float *__restrict__ post1,
    int32_t *__restrict__ route_record, int route, int B, int T, int S, int NB, int serial) {
    __shared__ float4 rows[2][NSEQ][kBlockMaxS / 4];      // posterior rows t-1 / t (entries >= S: -inf)
    __shared__ float upper_best[NSEQ][PQ - 1][kBlockMaxS];   // what the pieces pq >= 1 of the prev-states offer
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = wave % NB, pq = wave / NB;
    const int j = nb * 64 + lane;
    const bool live = j < S;
    const bool writer = pq == 0;
    const int lo = pq * L;
    if (blockIdx.x == 0 && tid == 0) *route_record = route;
    const float ninf = -__builtin_huge_valf();
    int n[NSEQ], nmax = 0;                                  // (0 past the batch: never active, nothing written)
    const float *o[NSEQ];
    float *h[NSEQ];
#pragma unroll
    for (int q = 0; q < NSEQ; ++q) {
        const int b = blockIdx.x * NSEQ + q;
        n[q] = 0;
        if (b < B) {
            n[q] = __builtin_amdgcn_readfirstlane(frames[b]);
            n[q] = n[q] < 1 ? 1 : (n[q] > T ? T : n[q]);
        }
        nmax = max(nmax, n[q]);
        const int bb = b < B ? b : B - 1;
        o[q] = obs + (size_t)bb * T * S + min(j, S - 1);
        h[q] = hist + (size_t)bb * T * S;
    }

    float row[L];                                          // trans[j][lo + e]
#pragma unroll
    for (int e = 0; e < L; ++e) row[e] = trans[(size_t)min(j, S - 1) * S + min(lo + e, S - 1)];   // (clamped addresses:
    asm volatile("" ::: "memory");                         //  every load unconditional and in flight before the first use)
    bool odd_matrix = nonfinite::odd(init[min(j, S - 1)]);     // (nonfinite.hpp: this route looks at the matrix itself)
#pragma unroll
    for (int e = 0; e < L; ++e) odd_matrix = odd_matrix || nonfinite::odd(row[e]);
    nonfinite::raise(odd_matrix, route_record + nonfinite::kMatrixWord, serial);
#pragma unroll
    for (int e = 0; e < L; ++e) row[e] = fminf(row[e], (live && lo + e < S) ? -ninf : ninf);          // the padding: -inf
    float p[NSEQ], cur[NSEQ][4];
    bool odd = false;                                      // a NaN / +inf posterior value was produced (nonfinite.hpp)
#pragma unroll
    for (int q = 0; q < NSEQ; ++q) {
        p[q] = o[q][0] + init[min(j, S - 1)];
        p[q] = live ? p[q] : ninf;
        odd = odd || (n[q] && nonfinite::odd(p[q]));
        if (writer) {
            reinterpret_cast<float *>(rows[0][q])[j] = p[q];
            if (live && n[q]) h[q][j] = p[q];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[q][k] = writer ? o[q][(size_t)min(1 + k, max(n[q], 1) - 1) * S] : 0.f;
    }
    __syncthreads();

    for (int t0 = 1; t0 < nmax; t0 += 4) {
        float nxt[NSEQ][4];
#pragma unroll
        for (int q = 0; q < NSEQ; ++q)                      // (the observation rows of the next four timesteps)
#pragma unroll
            for (int k = 0; k < 4; ++k) nxt[q][k] = writer ? o[q][(size_t)min(t0 + 4 + k, max(n[q], 1) - 1) * S] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (t0 + r < nmax) {                            // (uniform over the workgroup)
                float top[NSEQ];
#pragma unroll
                for (int q = 0; q < NSEQ; ++q) {
                    top[q] = ninf;
                    if (t0 + r < n[q]) {                    // (uniform over the workgroup)
                        const float4 *src = rows[(t0 + r - 1) & 1][q] + lo / 4;
                        float best[4] = {ninf, ninf, ninf, ninf};
#pragma unroll
                        for (int e = 0; e < L; e += 8) {    // (L = 48 or 64: whole groups of eight prev-states)
                            const float4 v0 = src[e / 4], v1 = src[e / 4 + 1];
                            best[0] = fmaxf(fmaxf(best[0], v0.x + row[e]), v1.x + row[e + 4]);
                            best[1] = fmaxf(fmaxf(best[1], v0.y + row[e + 1]), v1.y + row[e + 5]);
                            best[2] = fmaxf(fmaxf(best[2], v0.z + row[e + 2]), v1.z + row[e + 6]);
                            best[3] = fmaxf(fmaxf(best[3], v0.w + row[e + 3]), v1.w + row[e + 7]);
                        }
                        top[q] = fmaxf(fmaxf(best[0], best[1]), fmaxf(best[2], best[3]));
                        if (!writer && live) upper_best[q][pq - 1][j] = top[q];
                    }
                }
                __syncthreads();
                if (writer) {
#pragma unroll
                    for (int q = 0; q < NSEQ; ++q) {
                        if (t0 + r < n[q]) {
                            float best = top[q];
                            if (live) {
#pragma unroll
                                for (int u = 0; u < PQ - 1; ++u) best = fmaxf(best, upper_best[q][u][j]);
                            }
                            p[q] = live ? cur[q][r] + best : ninf;
                            odd = odd || nonfinite::odd(p[q]);
                            reinterpret_cast<float *>(rows[(t0 + r) & 1][q])[j] = p[q];
                            if (live) h[q][(size_t)(t0 + r) * S + j] = p[q];
                        }
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int q = 0; q < NSEQ; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[q][k] = nxt[q][k];
    }
#pragma unroll
    for (int q = 0; q < NSEQ; ++q) {
        const int b = blockIdx.x * NSEQ + q;
        if (writer && live && n[q]) (((n[q] - 1) & 1) ? post1 : post0)[(size_t)b * S + j] = p[q];
    }
    nonfinite::raise(odd, route_record + nonfinite::kAlarmWord, serial);
}

}  // namespace small
