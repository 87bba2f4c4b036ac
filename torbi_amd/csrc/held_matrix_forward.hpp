// held_matrix_forward.hpp -- the forward recurrence for a HANDFUL of sequences (B <= 16, S <= 4096) in ONE launch with
// the transition matrix held in registers across the whole chip.
//
// The reference decodes a batch of one with one block that walks the S x S matrix once per timestep
// (viterbi.cu:217-232); the per-timestep kernels here (step_rows*_kernel, small_batch_forward.hpp) spread a timestep over
// the chip but pay the gap between dependent launches -- 4.4 us at B = 1, S = 1440, where the arithmetic of a timestep is
// 2 M cells, a quarter of a microsecond of the chip's vector rate.  An 8.3 MB matrix does not fit a compute unit, but it
// fits the CHIP: ceil(S / 8) workgroups of 512 threads each hold 8 next-state rows of it in registers for the whole launch
// (8 x ceil(S / 512) values per thread; 24 at 1440 states; above 2048 states 16 rows per workgroup of 1024 threads, 64
// values per thread at 4096 states = the 64 MB matrix in the registers of 256 compute units) and the time loop runs
// inside the kernel.  Per timestep a
// workgroup needs the S posteriors of the step before, produced 8 apiece by all the others: they travel through a
// [2][B][S] buffer of 8-byte {value, timestep} words written and read with relaxed agent-scope 64-bit atomics (the "LL"
// hand-off of collective libraries: the tag arrives with the value in one store, so there is no flag round, no fence and
// no acknowledgement to wait for).  A consumer polls the S words it needs until every tag names the step it is waiting
// for; two parities suffice because a workgroup can only write step t + 2 after all others have published t + 1, i.e.
// consumed t.  Everything else is the reference's scan (viterbi.cpp:78-108): candidates fl(post[i] + trans[j][i]) in
// ascending i per thread, strict '>' (the first maximum wins), (value, index) pairs combined with the lower index on
// ties, posterior = fl(obs[t][j] + max), backpointer -> trellis; the final argmax and the chase stay in finalize_kernel.
//
// All workgroups (<= 256) must be resident at once -- one per compute unit of an MI355X, up to three at 1440 states.  That
// holds whenever the launch has the device to itself or shares it with a few of its kind; several such launches from
// different streams can each end up partly resident and wait for one another.  So every wait is bounded IN TIME (`wait_ticks`
// of the 100 MHz wall clock: 20 x T x 2.5 us, at least 2 ms -- a launch that is merely queued behind another kernel gets
// its compute units within that, a launch that waits for a workgroup that will never be resident does not hang a serving
// loop for longer), a workgroup that gives up says so in `control[1]` and goes on without waiting, and the launch is
// followed by `repair_kernel`, which does nothing when `control[1]` is 0 and otherwise decodes every sequence again with one
// workgroup each and no hand-offs (the reference's own kernel shape, viterbi.cu:48-130): slow, never wrong.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#ifndef HELD_FIRST_SLEEP
#define HELD_FIRST_SLEEP 16     // x 64 cycles before the first poll of a timestep (tools/held_probe.py, 1 x 500 x 1440,
                                // us per timestep: 0 -> 2.79, 8 -> 2.41, 12 -> 2.32, 16 -> 2.26, 20 -> 2.34, 32 -> 2.67)
#endif
#ifndef HELD_TWO_POINT
#define HELD_TWO_POINT 1        // two sequences in flight: the next one's words are asked for after the folds (1) / the scan (0)
#endif
#ifndef HELD_POLL_SLEEP
#define HELD_POLL_SLEEP 1       // x 64 cycles between polls
#endif

namespace held {

// Two shapes: up to 2048 states 8 next-states per workgroup of 512 threads (<= 256 workgroups, 4 prev-states per thread);
// up to 4096 states 16 next-states per workgroup of 1024 threads (<= 256 workgroups, 4 prev-states per thread, 64 matrix
// registers).  Either way one workgroup per compute unit of an MI355X, all resident at once.
constexpr int kMaxK = 4;              // prev-states per thread
constexpr int kMaxB = 16;
constexpr int kSmallS = 2048;
constexpr int kMaxS = 4096;
constexpr int kMaxRows = 16;

inline int rows_per_workgroup(int S) { return S <= kSmallS ? 8 : 16; }
inline int threads(int S) { return S <= kSmallS ? 512 : 1024; }                  // scanning threads
inline int block_threads(int S) { return S <= kSmallS ? 512 + 64 : 1024; }      // + the store wave where there is room
inline int workgroups(int S) { return (S + rows_per_workgroup(S) - 1) / rows_per_workgroup(S); }
inline bool supported(int B, int S, int cus) {
    return B >= 1 && B <= kMaxB && S >= 1 && S <= kMaxS && workgroups(S) <= (S <= kSmallS ? 2 * cus : cus);
}
inline size_t exchange_bytes(int B, int S) { return sizeof(unsigned long long) * 2 * (size_t)B * S; }

typedef unsigned long long u64;

// (selects, not branches: hipcc otherwise emits an exec-mask branch per comparison -- 47 in the time loop, 2.5 us per item)
__device__ __forceinline__ void better(float &v, int &i, float ov, int oi) {
    const bool take = (ov > v) | ((ov == v) & (oi < i));
    v = take ? ov : v;
    i = take ? oi : i;
}

// the other lane's value: DPP for partners inside a 16-lane row (W = 1, 2: quad permutes; 4: mirror the half row -- a
// matching between the lanes with bit 2 clear and set, all a fold needs; 8: rotate the row by 8), the LDS crossbar
// (ds_bpermute) across rows
template <int W>
__device__ __forceinline__ int partner(int x) {
    if constexpr (W == 1) return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
    else if constexpr (W == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
    else if constexpr (W == 4) return __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);   // row_half_mirror: l <-> 7 - l
    else if constexpr (W == 8) return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, true);
    else return __shfl_xor(x, W, 64);
}

// lanes l and l ^ W hold N rows each; afterwards each holds the N / 2 rows of its half (upper lanes the upper rows),
// combined over the pair
template <int W, int N, int R>
__device__ __forceinline__ void fold_pairs(float (&v)[R], int (&a)[R], bool upper) {
#pragma unroll
    for (int r = 0; r < N / 2; ++r) {
        const float sv = upper ? v[r] : v[r + N / 2];
        const int sa = upper ? a[r] : a[r + N / 2];
        const float ov = __int_as_float(partner<W>(__float_as_int(sv)));
        const int oa = partner<W>(sa);
        float kv = upper ? v[r + N / 2] : v[r];
        int ka = upper ? a[r + N / 2] : a[r];
        better(kv, ka, ov, oa);
        v[r] = kv;
        a[r] = ka;
    }
}

// The same fold for partners 32 or 16 lanes apart with gfx950's v_permlane32_swap / v_permlane16_swap: the instruction
// exchanges the upper half (odd 16-lane rows) of its first operand with the lower half (even rows) of its second, so with
// (lower row, upper row) as operands every lane ends with its own and its partner's value of the row it keeps -- no
// selects, no trip through the LDS crossbar.
template <int W, int N, int R>
__device__ __forceinline__ void fold_swap(float (&v)[R], int (&a)[R]) {
    static_assert(W == 32 || W == 16, "swap instructions exist for halves and 16-lane rows");
#pragma unroll
    for (int r = 0; r < N / 2; ++r) {
        const unsigned lo = __float_as_uint(v[r]), hi = __float_as_uint(v[r + N / 2]);
        const auto sv = W == 32 ? __builtin_amdgcn_permlane32_swap(lo, hi, false, false)
                                : __builtin_amdgcn_permlane16_swap(lo, hi, false, false);
        const auto sa = W == 32 ? __builtin_amdgcn_permlane32_swap((unsigned)a[r], (unsigned)a[r + N / 2], false, false)
                                : __builtin_amdgcn_permlane16_swap((unsigned)a[r], (unsigned)a[r + N / 2], false, false);
        float x = __uint_as_float(sv[0]);
        int ax = (int)sa[0];
        better(x, ax, __uint_as_float(sv[1]), (int)sa[1]);
        v[r] = x;
        a[r] = ax;
    }
}

// (value, index) of the best over the 4 lanes of a quad, in all of them
__device__ __forceinline__ void reduce_quad(float &v, int &a) {
    better(v, a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)),
           __builtin_amdgcn_update_dpp(0, a, 0xB1, 0xf, 0xf, true));
    better(v, a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true)),
           __builtin_amdgcn_update_dpp(0, a, 0x4E, 0xf, 0xf, true));
}

// (value, index) of the best over the 8 lanes that share bits 3..5 of the lane number, in all of them
__device__ __forceinline__ void reduce_eight(float &v, int &a) {
    better(v, a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)),
           __builtin_amdgcn_update_dpp(0, a, 0xB1, 0xf, 0xf, true));                       // quad_perm [1,0,3,2]
    better(v, a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true)),
           __builtin_amdgcn_update_dpp(0, a, 0x4E, 0xf, 0xf, true));                       // quad_perm [2,3,0,1]
    better(v, a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true)),
           __builtin_amdgcn_update_dpp(0, a, 0x141, 0xf, 0xf, true));                      // row_half_mirror
}

// Before the launch: row 0 of every sequence (viterbi.cpp:72-76) into post0 AND into parity 0 of the exchange as
// {value, timestep 0} words -- timestep 1 then reads its input like every other timestep --, parity 1 and the control
// words cleared (tags of an earlier decode with this workspace must not be taken for this one's).
__global__ __launch_bounds__(256) void prepare_kernel(const float *__restrict__ obs, const float *__restrict__ initial,
                                                      float *__restrict__ post0, u64 *__restrict__ xchg,
                                                      unsigned *__restrict__ control, int B, int T, int S) {
    const size_t n = (size_t)B * S;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / S);
        const int i = (int)(e - (size_t)b * S);
        const float v = obs[(size_t)b * T * S + i] + initial[i];
        post0[e] = v;
        xchg[e] = (u64)__float_as_uint(v);
        xchg[n + e] = 0ull;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) control[threadIdx.x] = 0u;
}

// grid = workgroups(S), block = block_threads(S), after prepare_kernel.
// K prev-states per thread, kRows next-states per workgroup (8 or 16), kThreads scanning threads.  With STORE_WAVE the
// block carries one more wave that does nothing but combine the scanning waves' results and store them: on gfx9 stores
// count on the same per-wave counter as loads and a wave with both outstanding can only wait for everything, so a
// scanning wave that had stored would sit out the acknowledgement of a write-through store (~0.5 us) at its next poll.
// The store wave never waits for memory: its observations reach it through the LDS from wave 0, which asks for them
// together with its words.  (1024 scanning threads leave no room for a further wave: there wave 0 stores.)
template <int K, int kRows, int kThreads, bool STORE_WAVE>
__global__ __launch_bounds__(kThreads + (STORE_WAVE ? 64 : 0)) void held_forward_kernel(
    const float *__restrict__ obs, const int32_t *__restrict__ frames, const float *__restrict__ trans,
    float *__restrict__ post0, float *__restrict__ post1, int32_t *__restrict__ trellis, u64 *__restrict__ xchg,
    unsigned *__restrict__ control, int B, int T, int S, unsigned long long wait_ticks) {
    constexpr int kWaves = kThreads / 64;                       // scanning waves
    static_assert(kRows == 8 || kRows == 16, "folds are written for 8 and 16 rows");
    __shared__ float sv[2][kWaves][kRows];
    __shared__ int sa[2][kWaves][kRows];
    __shared__ float sob[2][kRows];
    __shared__ int sframes[kMaxB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool scans = wave < kWaves;
    const bool stores = (STORE_WAVE ? wave == kWaves : wave == 0) && lane < kRows && blockIdx.x * kRows + lane < S;
    const bool observes = wave == 0 && lane < kRows && blockIdx.x * kRows + lane < S;     // asks for the observations
    const int j0 = blockIdx.x * kRows;
    const int jmine = j0 + lane;
    const bool last_valid = tid + kThreads * (K - 1) < S;       // only the last of a thread's prev-states can lie beyond S

    // this workgroup's rows of the matrix, for the whole launch
    float tr[kRows][K];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int j = j0 + r < S ? j0 + r : S - 1;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = tid + kThreads * k;
            tr[r][k] = (scans && (k < K - 1 || last_valid)) ? trans[(size_t)j * S + i] : 0.0f;
        }
    }
    if (tid < kMaxB) {
        int f = tid < B ? frames[tid] : 1;
        sframes[tid] = f < 1 ? 1 : (f > T ? T : f);
    }
    __syncthreads();
    int longest = 1;
    for (int b = 0; b < B; ++b) longest = max(longest, sframes[b]);
    bool gave_up = false;
    int round = 0;                                              // work items done so far

    // What a work item (timestep t of sequence b) reads: the words of posterior row t-1 this thread scans and, in wave 0,
    // the observations of the output rows -- requested together, so that whoever waits for the words has the observation
    // too and nothing later in the item waits on the vector-memory counter (loads return in order; a wait the compiler
    // places behind a conditional request is a wait for that request: 1 us).
    struct Inputs { u64 w[K]; float ob; };
    auto request = [&](int t, int b, Inputs &in) {
        const u64 *src = xchg + ((size_t)((t - 1) & 1) * B + b) * S;
        in.ob = observes ? obs[((size_t)b * T + t) * S + jmine] : 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k)
            in.w[k] = (k < K - 1 || last_valid)
                          ? __hip_atomic_load(src + tid + kThreads * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : (u64)(unsigned)(t - 1) << 32;
    };
    auto complete = [&](const Inputs &in, int t) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < K; ++k) ok = ok && (unsigned)(in.w[k] >> 32) == (unsigned)(t - 1);
        return __builtin_amdgcn_ballot_w64(!ok) == 0;
    };
    // work items in order: every sequence that has not ended, timestep by timestep (uniform over the whole grid); lanes
    // 0..B-1 of every wave keep the lengths, so "which sequences are alive at t" is one ballot
    const int my_frames = sframes[lane & (kMaxB - 1)];
    auto alive = [&](int t) -> unsigned {
        return (unsigned)__builtin_amdgcn_ballot_w64(lane < B && my_frames > t);
    };
    auto advance = [&](int &t, int &b) {
        unsigned m = alive(t);
        if (b >= 0) m &= ~((2u << b) - 1u);                      // the alive sequences behind b
        while (m == 0u && t < longest) m = alive(++t);
        b = m ? __builtin_ctz(m) : 0;
    };
    int t = 1, b = -1;
    advance(t, b);

    if (scans) {
        Inputs ahead = {};        // the NEXT item's inputs while they are in flight
        bool requested = false;
        if (t < longest) {        // (row 0 is in place: the first item's inputs can be asked for at once)
            request(t, b, ahead);
            requested = true;
        }
        while (t < longest) {
            int nt = t, nb = b;
            advance(nt, nb);
            const int others = __builtin_popcount(alive(t)) - 1;
            // The next item's inputs can be asked for once its previous row has had time to become visible (~1 us after
            // that grid-wide store; asked for earlier, the poll fails and the next one waits behind it):
            //   three or more sequences in flight -- it was stored at least one item ago: ask at the top of this item;
            //   two -- it was stored as this item began: ask after this item's scan / folds (HELD_TWO_POINT);
            //   one -- the next item is this sequence again: sleep, then ask (below).
            // (Measured and dropped: a second poll in flight half a round trip behind the first -- 3.1 against 2.8 us
            // per timestep; the hand-off is paid in the consumer's own memory queue, every extra poll lengthens it.)
            const bool prefetch = nt < longest && nb != b;
            Inputs mine = {};
            bool ok = false;
            if (requested) {
                mine = ahead;                                    // (waits for them)
                ok = complete(mine, t);
            }
            requested = false;
            if (prefetch && others >= 2) {
                request(nt, nb, ahead);
                requested = true;
            }
            if (!ok) {
                const unsigned long long waiting_since = wall_clock64();     // (100 MHz, the same on every compute unit)
                if (others == 0) __builtin_amdgcn_s_sleep(HELD_FIRST_SLEEP);
                for (;;) {
                    request(t, b, mine);
                    if (complete(mine, t)) break;
                    // (once given up, never wait again)
                    if (gave_up || wall_clock64() - waiting_since >= wait_ticks) { gave_up = true; break; }
                    __builtin_amdgcn_s_sleep(HELD_POLL_SLEEP);
                }
            }
            float p[K];
#pragma unroll
            for (int k = 0; k < K; ++k) p[k] = __uint_as_float((unsigned)mine.w[k]);
            // the reference's scan over this thread's prev-states, for each of the rows
            float v[kRows];
            int a[kRows];
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                v[r] = p[0] + tr[r][0];                          // (K == 1: a thread beyond S holds no candidate)
                a[r] = tid;
                if (K == 1 && !last_valid) { v[r] = -INFINITY; a[r] = 0x7fffffff; }
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    const float c = p[k] + tr[r][k];
                    const bool take = (c > v[r]) & (k < K - 1 || last_valid);
                    v[r] = take ? c : v[r];
                    a[r] = take ? tid + kThreads * k : a[r];
                }
            }
#if HELD_TWO_POINT == 0
            if (prefetch && !requested) {
                request(nt, nb, ahead);
                requested = true;
            }
#endif
            // rows x 64 lanes -> one row per lane (halving folds), then over the lanes that share a row
            int row;
            bool keeper;
            if constexpr (kRows == 8) {
                fold_swap<32, 8, kRows>(v, a);
                fold_swap<16, 4, kRows>(v, a);
                fold_pairs<8, 2, kRows>(v, a, (lane & 8) != 0);
                reduce_eight(v[0], a[0]);
                row = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
                keeper = (lane & 7) == 0;
            } else {
                fold_swap<32, 16, kRows>(v, a);
                fold_swap<16, 8, kRows>(v, a);
                fold_pairs<8, 4, kRows>(v, a, (lane & 8) != 0);
                fold_pairs<4, 2, kRows>(v, a, (lane & 4) != 0);
                reduce_quad(v[0], a[0]);
                row = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
                keeper = (lane & 3) == 0;
            }
#if HELD_TWO_POINT == 1
            if (prefetch && !requested) {
                request(nt, nb, ahead);
                requested = true;
            }
#endif
            const int slot = round & 1;
            if (keeper) {
                sv[slot][wave][row] = v[0];
                sa[slot][wave][row] = a[0];
            }
            if (observes) sob[slot][lane] = mine.ob;
            __syncthreads();
            if (!STORE_WAVE && stores) {
                float bv = sv[slot][0][lane];
                int ba = sa[slot][0][lane];
#pragma unroll
                for (int w = 1; w < kWaves; ++w) better(bv, ba, sv[slot][w][lane], sa[slot][w][lane]);
                const float out = sob[slot][lane] + bv;
                const u64 word = ((u64)(unsigned)t << 32) | __float_as_uint(out);
                __hip_atomic_store(xchg + ((size_t)(t & 1) * B + b) * S + jmine, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                trellis[((size_t)b * T + t) * S + jmine] = ba;
                if (t == sframes[b] - 1) ((t & 1) ? post1 : post0)[(size_t)b * S + jmine] = out;
            }
            ++round;
            t = nt;
            b = nb;
        }
        if (gave_up && lane == 0) atomicAdd(control + 1, 1u);
    } else {
        // the store wave: item after item, wait for the scanning waves, combine, store
        while (t < longest) {
            const int slot = round & 1;
            __syncthreads();
            if (stores) {
                float bv = sv[slot][0][lane];
                int ba = sa[slot][0][lane];
#pragma unroll
                for (int w = 1; w < kWaves; ++w) better(bv, ba, sv[slot][w][lane], sa[slot][w][lane]);
                const float out = sob[slot][lane] + bv;
                const u64 word = ((u64)(unsigned)t << 32) | __float_as_uint(out);
                __hip_atomic_store(xchg + ((size_t)(t & 1) * B + b) * S + jmine, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                trellis[((size_t)b * T + t) * S + jmine] = ba;
                if (t == sframes[b] - 1) ((t & 1) ? post1 : post0)[(size_t)b * S + jmine] = out;
            }
            ++round;
            advance(t, b);
        }
    }
}

// The safety net behind a held launch (see the top of the file): one workgroup per sequence, posterior rows ping-pong in
// the LDS, one wave per next-state and round, lanes over the prev-states in ascending stripes (the reference scan:
// strict '>', lower index on ties).  grid = B, block = 1024, dynamic LDS = 2 * S floats.
__global__ __launch_bounds__(1024) void repair_kernel(const float *__restrict__ obs, const int32_t *__restrict__ frames,
                                                      const float *__restrict__ trans, const float *__restrict__ initial,
                                                      float *__restrict__ post0, float *__restrict__ post1,
                                                      int32_t *__restrict__ trellis, const unsigned *__restrict__ control,
                                                      int B, int T, int S) {
    if (control[1] == 0u) return;                               // every workgroup of the held launch completed
    extern __shared__ __attribute__((aligned(16))) float rows[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int f = frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    for (int i = tid; i < S; i += 1024) {
        const float v = obs[(size_t)b * T * S + i] + initial[i];
        rows[i] = v;
        post0[(size_t)b * S + i] = v;
    }
    __syncthreads();
    for (int t = 1; t < f; ++t) {
        const float *cur = rows + (size_t)((t - 1) & 1) * S;
        float *nxt = rows + (size_t)(t & 1) * S;
        for (int j = wave; j < S; j += 16) {
            const float *row = trans + (size_t)j * S;
            float v = -INFINITY;
            int a = 0x7fffffff;
            for (int i = lane; i < S; i += 64) {
                const float c = cur[i] + row[i];
                const bool take = (c > v) | (a == 0x7fffffff);   // the first candidate is taken whatever it is
                v = take ? c : v;
                a = take ? i : a;
            }
#pragma unroll
            for (int w = 32; w > 0; w >>= 1) {
                const float ov = __shfl_xor(v, w, 64);
                const int oa = __shfl_xor(a, w, 64);
                better(v, a, ov, oa);
            }
            if (lane == 0) {
                const float out = obs[((size_t)b * T + t) * S + j] + v;
                nxt[j] = out;
                trellis[((size_t)b * T + t) * S + j] = a;
                if (t == f - 1) ((t & 1) ? post1 : post0)[(size_t)b * S + j] = out;
            }
        }
        __syncthreads();
    }
}

}  // namespace held
