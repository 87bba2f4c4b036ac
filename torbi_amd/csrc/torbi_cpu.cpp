// torbi_cpu.cpp -- host twin of the MI355X decoder behind include/torbi_cpu.h.
//
// The operator the reference registers for the CPU key (viterbi_decode_cpu, torbi/csrc/viterbi.cpp:182-234), written
// the way the HIP side is: the forward pass keeps VALUES only,
//     post'[j] = fl(obs[t,j] + max_i fl(post[i] + trans[j,i]))                       (viterbi.cpp:81-104)
// as a (max,+) row product vectorised over prev-states with a block of items sharing every transition row, the
// posterior rows are kept as the (B,T,S) history, and the backpointer -- lowest prev-state attaining the maximum
// (viterbi.cpp:94-100) -- is recomputed only for the states on the decoded path (viterbi.cpp:140-160 reads exactly
// those).  Same two roundings per cell in the same order, max is exact, so indices are bit-identical to the reference
// operator.  NaN and +inf inputs included: the reference is deterministic there (a NaN candidate at prev-state 0 is never
// replaced, one elsewhere never wins, viterbi.cpp:94-100; the final state is ATen's argmax: the first NaN, :218), a vectorised
// maximum is not, so every posterior value produced is looked at (NaN / +inf observations show there) and an item that met
// one -- every item when the matrix or the initial vector hold one -- is decoded again in the reference's own order of
// evaluation (faithful_item below; the HIP side does the same, csrc/nonfinite.hpp).  Not the test oracle (oracle/viterbi_oracle.c restates the reference's own loop
// structure) and not a fallback of the HIP path: torbi_amd calls it only where a caller asks for the CPU (gpu=None).
#include "torbi_cpu.h"
#include "file_rows.hpp"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <omp.h>
#include <vector>

namespace {

#ifndef TORBI_CPU_BLOCK
#define TORBI_CPU_BLOCK 8
#endif
constexpr int kBlock = TORBI_CPU_BLOCK;   // items that share a pass over the transition matrix (8: their posterior rows stay
                                          // in the L1/L2 of a core; 16 measured 1.6x slower on an EPYC 9xx5)
constexpr int kLanes = 16;     // floats per accumulator (one 512-bit register; two 256-bit ones)
constexpr float kNegInf = -std::numeric_limits<float>::infinity();

inline int clamp_frames(int f, int T) { return f < 1 ? 1 : (f > T ? T : f); }

// out[b][j] = obs[b][j] + max_i (post[b][i] + trans[j][i]) for next-states [j0, j1) and NB items.  The item loop is
// innermost: every 16-float piece of a transition row is loaded once and added to NB posterior rows.
inline bool odd(float x) { return !(x <= std::numeric_limits<float>::max()); }        // NaN or +inf (-inf is ordinary)

template <int NB>
__attribute__((target_clones("avx512f", "avx2", "default")))
void step_rows(const float *const *post, const float *const *obs, float *const *out, const float *trans, int S, int j0,
               int j1, unsigned char *const *flag) {
    const int whole = S / kLanes * kLanes;
    for (int j = j0; j < j1; ++j) {
        const float *tr = trans + (size_t)j * S;
        float acc[NB][kLanes];
        for (int b = 0; b < NB; ++b)
            for (int l = 0; l < kLanes; ++l) acc[b][l] = kNegInf;
        for (int i = 0; i < whole; i += kLanes) {
            for (int b = 0; b < NB; ++b) {
                const float *p = post[b] + i;
#pragma omp simd
                for (int l = 0; l < kLanes; ++l) {
                    const float c = p[l] + tr[i + l];
                    acc[b][l] = c > acc[b][l] ? c : acc[b][l];
                }
            }
        }
        for (int b = 0; b < NB; ++b) {
            float m = kNegInf;
            for (int l = 0; l < kLanes; ++l) m = acc[b][l] > m ? acc[b][l] : m;
            for (int i = whole; i < S; ++i) {
                const float c = post[b][i] + tr[i];
                m = c > m ? c : m;
            }
            const float v = obs[b][j] + m;
            out[b][j] = v;
            if (odd(v)) __atomic_store_n(flag[b], (unsigned char)1, __ATOMIC_RELAXED);
        }
    }
}

void step_rows_any(int nb, const float *const *post, const float *const *obs, float *const *out, const float *trans, int S,
                   int j0, int j1, unsigned char *const *flag) {
    switch (nb) {
        case 8: step_rows<8>(post, obs, out, trans, S, j0, j1, flag); break;
        case 7: step_rows<7>(post, obs, out, trans, S, j0, j1, flag); break;
        case 6: step_rows<6>(post, obs, out, trans, S, j0, j1, flag); break;
        case 5: step_rows<5>(post, obs, out, trans, S, j0, j1, flag); break;
        case 4: step_rows<4>(post, obs, out, trans, S, j0, j1, flag); break;
        case 3: step_rows<3>(post, obs, out, trans, S, j0, j1, flag); break;
        case 2: step_rows<2>(post, obs, out, trans, S, j0, j1, flag); break;
        case 1: step_rows<1>(post, obs, out, trans, S, j0, j1, flag); break;
        default: break;
    }
}

// One item exactly as the reference decodes it (viterbi.cpp:65-108, 140-160, 218-221): scan order, strict '>', int32 trellis,
// ATen's argmax (the first NaN of the final row, else its first maximum).  `trellis`: [T][S] int32 (the item's history region).
void faithful_item(const float *obs, int f, const float *trans, const float *initial, int32_t *trellis, int32_t *out, int T,
                   int S) {
    std::vector<float> rows((size_t)2 * S);
    float *cur = rows.data(), *nxt = cur + S;
    for (int i = 0; i < S; ++i) cur[i] = obs[i] + initial[i];
    for (int t = 1; t < f; ++t) {
        for (int j = 0; j < S; ++j) {
            const float *tr = trans + (size_t)j * S;
            float best = cur[0] + tr[0];
            int arg = 0;
            for (int i = 1; i < S; ++i) {
                const float c = cur[i] + tr[i];
                if (c > best) { best = c; arg = i; }
            }
            trellis[(size_t)t * S + j] = arg;
            nxt[j] = obs[(size_t)t * S + j] + best;
        }
        std::swap(cur, nxt);
    }
    int arg = 0;
    float best = cur[0];
    for (int i = 1; i < S; ++i)
        if (cur[i] > best || (cur[i] != cur[i] && best == best)) { best = cur[i]; arg = i; }
    for (int t = f - 1; t < T; ++t) out[t] = arg;
    int index = arg;
    for (int t = f - 1; t >= 1; --t) {
        index = trellis[(size_t)t * S + index];
        out[t - 1] = index;
    }
}

// lowest index of the maximum of v[0..n)   (first-max semantics of the reference's argmax, viterbi.cpp:218)
inline int first_argmax(const float *v, int n) {
    int k = 0;
    float m = v[0];
    for (int i = 1; i < n; ++i)
        if (v[i] > m) { m = v[i]; k = i; }
    return k;
}

// the backpointer of state j at a timestep whose previous posterior row is `prev`: lowest i attaining
// max_i (prev[i] + trans[j][i]); a running maximum from i = 0 replaced on strict > (viterbi.cpp:94-100)
__attribute__((target_clones("avx512f", "avx2", "default")))
int backpointer(const float *prev, const float *tr, int S) {
    float m = kNegInf;
#pragma omp simd reduction(max : m)
    for (int i = 0; i < S; ++i) {
        const float c = prev[i] + tr[i];
        m = c > m ? c : m;
    }
    for (int i = 0; i < S; ++i)
        if (prev[i] + tr[i] == m) return i;
    return 0;       // every candidate is NaN-free -inf: the reference's running maximum never leaves index 0
}

void backtrace_item(const float *hist, const float *trans, int32_t *out, int f, int T, int S) {
    int j = first_argmax(hist + (size_t)(f - 1) * S, S);
    for (int t = f - 1; t < T; ++t) out[t] = j;              // every position t >= frames-1 (viterbi.cpp:219-221)
    for (int t = f - 1; t >= 1; --t) {
        j = backpointer(hist + (size_t)(t - 1) * S, trans + (size_t)j * S, S);
        out[t - 1] = j;
    }
}

// barrier of one team of threads inside a flat parallel region (an `omp barrier` would stop every team)
struct alignas(64) TeamBarrier {
    std::atomic<unsigned> arrived{0}, phase{0};
    void wait(int size, unsigned &my_phase) {
        if (size <= 1) return;
        const unsigned next = ++my_phase;
        if (arrived.fetch_add(1u, std::memory_order_acq_rel) + 1u == (unsigned)size) {
            arrived.store(0u, std::memory_order_relaxed);
            phase.store(next, std::memory_order_release);
        } else {
            int spins = 0;
            while (phase.load(std::memory_order_acquire) != next)
                if (++spins > 4096) { std::this_thread::yield(); spins = 0; }
        }
    }
};

struct Block {
    int first, count;       // items [first, first + count)
    int longest;            // frames of its longest item
};

// the forward pass of one block over next-states [j0, j1) of timestep t (items that have ended are left out)
inline void block_step(const Block &blk, const int *frames, const float *obs, float *hist, const float *trans, int T, int S,
                       int t, int j0, int j1, unsigned char *flags) {
    const float *post[kBlock], *ob[kBlock];
    float *out[kBlock];
    unsigned char *flag[kBlock];
    int nb = 0;
    for (int k = 0; k < blk.count; ++k) {
        const int b = blk.first + k;
        if (t >= frames[b]) continue;
        const size_t base = (size_t)b * T * S;
        post[nb] = hist + base + (size_t)(t - 1) * S;
        ob[nb] = obs + base + (size_t)t * S;
        out[nb] = hist + base + (size_t)t * S;
        flag[nb] = flags + b;
        ++nb;
    }
    step_rows_any(nb, post, ob, out, trans, S, j0, j1, flag);
}

}  // namespace

extern "C" {

int torbi_cpu_abi_version(void) { return TORBI_CPU_ABI_VERSION; }

int torbi_cpu_viterbi_decode(const float *observation, const int32_t *batch_frames, const float *transition,
                             const float *initial, int32_t *indices_out, int B, int T, int S, int num_threads) {
    if (B < 0 || T < 1 || S < 1) return TORBI_CPU_EINVAL;
    if (B == 0) return TORBI_CPU_OK;
    if (!observation || !batch_frames || !transition || !initial || !indices_out) return TORBI_CPU_EINVAL;
    // default: at most 32 threads -- the transition matrix is streamed once per item block and timestep, and beyond that the
    // shared caches are the limit (measured on a 256-thread EPYC: 512 items 428 K timesteps/s at 32 threads, 150 K at 256)
    const int threads = num_threads > 0 ? num_threads : std::min(omp_get_max_threads(), 32);

    float *hist = static_cast<float *>(std::aligned_alloc(64, ((size_t)B * T * S * sizeof(float) + 63) / 64 * 64));
    if (!hist) return TORBI_CPU_ENOMEM;
    std::vector<int> frames((size_t)B);
    for (int b = 0; b < B; ++b) frames[b] = clamp_frames(batch_frames[b], T);

    std::vector<Block> blocks;
    for (int first = 0; first < B; first += kBlock) {
        Block blk{first, std::min(kBlock, B - first), 0};
        for (int k = 0; k < blk.count; ++k) blk.longest = std::max(blk.longest, frames[first + k]);
        blocks.push_back(blk);
    }
    const int nblocks = (int)blocks.size();

    // items that met a NaN / +inf (in a posterior value they produced: the observations show there; or in the matrix / the
    // initial vector: every item) are decoded again in the reference's order once everything else is done
    std::vector<unsigned char> flagged((size_t)B, 0);
    unsigned char *const flags = flagged.data();
    bool odd_matrix = false;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(|| : odd_matrix)
    for (int j = 0; j < S; ++j) {
        const float *tr = transition + (size_t)j * S;
        bool seen = odd(initial[j]);
        for (int i = 0; i < S; ++i) seen = seen || odd(tr[i]);
        odd_matrix = odd_matrix || seen;
    }

    // t = 0: post = obs[0] + initial                                             (viterbi.cpp:72-76)
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int b = 0; b < B; ++b) {
        const float *o = observation + (size_t)b * T * S;
        float *h = hist + (size_t)b * T * S;
        bool seen = odd_matrix;
        for (int i = 0; i < S; ++i) {
            h[i] = o[i] + initial[i];
            seen = seen || odd(h[i]);
        }
        if (seen) flags[b] = 1;
    }

    // item blocks are dealt to teams of threads; a team of one takes whole blocks through all their timesteps without
    // any barrier, a larger team splits the next-states of every timestep (one team barrier per timestep)
    const int teams = std::min(nblocks, threads);
    const int team_size = std::max(1, threads / teams);
    if (team_size == 1) {
#pragma omp parallel for num_threads(teams) schedule(dynamic, 1)
        for (int n = 0; n < nblocks; ++n) {
            const Block &blk = blocks[n];
            for (int t = 1; t < blk.longest; ++t)
                block_step(blk, frames.data(), observation, hist, transition, T, S, t, 0, S, flags);
            for (int k = 0; k < blk.count; ++k) {
                const int b = blk.first + k;
                backtrace_item(hist + (size_t)b * T * S, transition, indices_out + (size_t)b * T, frames[b], T, S);
            }
        }
    } else {
        // ONE flat parallel region (nothing process-wide is touched: no nested parallelism, no omp_set_* call): thread id ->
        // (team, place in the team); a team takes item blocks team, team + teams, ... and meets at its own barrier
        std::vector<TeamBarrier> barriers((size_t)teams);
#pragma omp parallel num_threads(teams * team_size)
        {
            const int nth = omp_get_num_threads(), id = omp_get_thread_num();
            const int nteams = std::min(teams, nth), size = nth / nteams;      // what the runtime actually gave us
            const int team = id / size, me = id % size;
            if (team < nteams) {
                TeamBarrier &bar = barriers[(size_t)team];
                unsigned phase = 0;
                const int j0 = (int)((long long)S * me / size), j1 = (int)((long long)S * (me + 1) / size);
                for (int n = team; n < nblocks; n += nteams) {
                    const Block &blk = blocks[n];
                    for (int t = 1; t < blk.longest; ++t) {
                        block_step(blk, frames.data(), observation, hist, transition, T, S, t, j0, j1, flags);
                        bar.wait(size, phase);
                    }
                    for (int k = me; k < blk.count; k += size) {
                        const int b = blk.first + k;
                        backtrace_item(hist + (size_t)b * T * S, transition, indices_out + (size_t)b * T, frames[b], T, S);
                    }
                }
            }
        }
    }
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        if (flags[b])
            faithful_item(observation + (size_t)b * T * S, frames[b], transition, initial,
                          reinterpret_cast<int32_t *>(hist + (size_t)b * T * S), indices_out + (size_t)b * T, T, S);
    std::free(hist);
    return TORBI_CPU_OK;
}

int torbi_cpu_read_rows(const int *fds, const int64_t *offsets, const int64_t *bytes, void *const *rows,
                        const int64_t *zero_bytes, int count, int threads, int *error_out) {
    if (count < 0 || threads < 1) return TORBI_CPU_EINVAL;
    if (count == 0) return TORBI_CPU_OK;
    if (!fds || !offsets || !bytes || !rows || !zero_bytes) return TORBI_CPU_EINVAL;
    for (int k = 0; k < count; ++k)
        if (fds[k] < 0 || offsets[k] < 0 || bytes[k] < 0 || zero_bytes[k] < 0 || (!rows[k] && bytes[k] + zero_bytes[k] > 0))
            return TORBI_CPU_EINVAL;
    const int rc = filerows::read_rows(fds, offsets, bytes, rows, zero_bytes, count, threads, error_out);
    return rc == 0 ? TORBI_CPU_OK : TORBI_CPU_EIO_BASE + rc + 1;      // -(100 + index)
}

int torbi_cpu_write_files(const char *const *paths, const void *const *data, const int64_t *bytes, int count, int threads,
                          int *error_out) {
    if (count < 0 || threads < 1) return TORBI_CPU_EINVAL;
    if (count == 0) return TORBI_CPU_OK;
    if (!paths || !data || !bytes) return TORBI_CPU_EINVAL;
    for (int k = 0; k < count; ++k)
        if (!paths[k] || bytes[k] < 0 || (!data[k] && bytes[k] > 0)) return TORBI_CPU_EINVAL;
    const int rc = filerows::write_files(paths, data, bytes, count, threads, error_out);
    return rc == 0 ? TORBI_CPU_OK : TORBI_CPU_EIO_BASE + rc + 1;      // -(100 + index)
}

int torbi_cpu_open_heads(const char *const *paths, int count, int threads, int head_bytes, int *fds_out,
                         unsigned char *heads_out, int *lengths_out, int *error_out) {
    if (count < 0 || threads < 1 || head_bytes < 1) return TORBI_CPU_EINVAL;
    if (count == 0) return TORBI_CPU_OK;
    if (!paths || !fds_out || !heads_out || !lengths_out) return TORBI_CPU_EINVAL;
    for (int k = 0; k < count; ++k)
        if (!paths[k]) return TORBI_CPU_EINVAL;
    const int rc = filerows::open_heads(paths, count, threads, head_bytes, fds_out, heads_out, lengths_out, error_out);
    return rc == 0 ? TORBI_CPU_OK : TORBI_CPU_EIO_BASE + rc + 1;      // -(100 + index)
}

}  // extern "C"
