// band_forward.hpp -- the forward recurrence for BANDED transition matrices, time loop inside one launch, every finite
// cell evaluated, no cell outside the band touched.
//
// The reference's own workload is banded: torbi.evaluate builds the pitch transition clip(w - |x - y|, 0) with w = 87.2 of
// 1440 bins (torbi/evaluate/core.py:24-33), so a next-state j has at most 175 finite candidates, j - 87 <= i <= j + 87,
// and every other candidate of viterbi.cpp:81-104 is fl(post[i] + (-inf)) = -inf, which never wins the strict '>' of
// viterbi.cpp:94-100.  The recurrence over the band alone therefore yields the reference's posterior rows bit for bit:
//     post'[j] = fl( obs[t][j] + max_{j - hl <= i <= j + hr} fl( post[i] + trans[j][i] ) )
// (max is exact and order independent; rows whose finite range is narrower than the band carry their -inf entries along).
// The history of posterior rows goes to HBM and the backtrace recomputes the first argmax along the decoded path inside
// the band (group_backtrace_band_kernel below), as lazy_backtrace.hpp does for whole rows.
//
// Shape of the work.  A tile is 16 items; R workgroups (members) share it, member r owns next-states [r n, (r + 1) n),
// n = ceil4(S / R) <= 192.  What a member keeps in its LDS for the whole launch:
//     Tq  [dquad q][jg][4 diagonals][4 next]   its slab of the band, diagonal-major: diagonal dd = i - j + hl, four
//                                              diagonals to a "dquad", next-states in groups of four (jg)      126.7 KB
//     W   [4 item groups][rows][4 items]       the window of the previous posterior row it reads: its own n rows and hl / hr
//                                              rows of its neighbours either side (the halo)                    23.6 KB
//     M   [16 (next, item) of a lane][lanes]   partial maxima of the waves that share outputs (ds_max_f32)       11.5 KB
// (bytes at 1440 states, R = 8, hl = hr = 87: 162 112 of 163 840).  A lane owns 4 next-states x 4 items for a run of
// diagonals: per dquad it reads 4 x 16 B of Tq and 4 new rows x 16 B of W (the window slides: a row is read once per lane
// and used by up to 4 diagonals x 4 next-states) for 64 cells = 64 v_add_f32 + 32 v_max3_f32: 1 byte of LDS per cell at 256
// B/clk against 4 issue cycles per cell-wave -- the vector ALU binds, the LDS runs at half its rate.  A wave = 16 groups of next-states x the 4
// item groups; the waves of a 64-next block split its diagonals four ways and merge through M.
//
// Halo.  Per timestep a member needs hl + hr rows x 16 items from its two neighbours.  They travel as self-validating
// 16-byte granules {v0, tag, v1, tag} (tag = timestep + 1; write-through stores, L1-bypassing loads; the granule's two
// 8-byte halves are each {value, tag}: MI355X_MICROARCH.md "R2's granule") through a per-tile exchange buffer by timestep
// parity -- one trip, no flag, no fence.  Each wave evaluates the dquads that touch only its member's own rows FIRST (50-66 %
// of its work), asks for the halo granules half-way through them, and only then waits: the hand-off (~1.5-2 us) hides
// behind ~2 us of own-row work.  Membership is by arrival ticket as in the cluster form of resident_forward.hpp; every
// wait is bounded (Exchange::wait_ticks), a member that gives up flags its tile and band_repair_kernel decodes it again
// without hand-offs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "resident_forward.hpp"
#include "lazy_backtrace.hpp"

namespace band {

using resident::Batch;
using resident::Group;
using resident::buffer_of;
using resident::v4u;

constexpr int kNI = 16;                  // items per tile
constexpr int kMaxR = 16;                // members per tile
constexpr int kMaxBlocks = 3;            // 64-next blocks per member (n <= 192)
constexpr int kMaxWaves = 4 * kMaxBlocks;
constexpr int kMaxRounds = 4;            // halo granules a thread asks for per timestep
constexpr int kLdsBytes = 160 * 1024;
constexpr int kMaxWindow = 512;          // the backtrace's window of prev-states (two float4 per lane)

struct Plan {
    int S, hl, hr;           // trans[j][i] is -inf unless j - hl <= i <= j + hr (the caller's promise)
    int R;                   // members per tile
    int n_own;               // next-states per member (multiple of 4)
    int n_jg;                // n_own / 4
    int nblk;                // 64-next blocks per member
    int waves;               // 4 per block
    int Dq;                  // dquads: ceil((hl + hr + 1) / 4)
    int w_rows;              // rows of the window: n_own + 4 Dq - 1 (row w holds state j0 - hl + w)
    int ig_stride;           // floats between the item groups' windows (== 4 mod 64: conflict-free 16-byte reads)
    int td_off, w_off, m_off, misc_off, lds_bytes;
    int rounds;              // halo granules per thread and timestep
    // per wave, in dquads: [0],[1] and [2],[3] the ones that read only the member's own rows (the halo granules are asked
    // for between the two runs), [4],[5] and [6],[7] the ones that read halo rows
    short seg[kMaxWaves][8];
};

struct Exchange {
    char *xchg[resident::kMaxBatches];   // per batch: [tiles][2 parities][4 item groups][S][2][16 bytes]
    unsigned *control;                   // [0] tickets drawn (zeroed before the launch)
    unsigned *failed;                    // [tiles of the group] set by a member that gave up waiting (zeroed before the launch)
    unsigned long long wait_ticks;       // budget of one wait (100 MHz ticks)
};

__host__ __device__ inline size_t xchg_tile_bytes(int S) { return (size_t)2 * 4 * (size_t)S * 32; }
__host__ __device__ inline size_t xchg_bytes(int B, int S) { return (size_t)((B + kNI - 1) / kNI) * xchg_tile_bytes(S); }

// The plan for `tiles` tiles on `cus` compute units, or false when the band kernel does not cover the shape.
inline bool make_plan(int S, int hl, int hr, int tiles, int cus, Plan &pl) {
    if (S < 64 || S % 4 != 0 || hl < 0 || hr < 0 || hl >= S || hr >= S || tiles < 1) return false;
    if (hl + hr + 4 > kMaxWindow) return false;
    const int Dq = (hl + hr + 1 + 3) / 4;
    const int want = std::max(1, std::min(kMaxR, cus / tiles));
    int best = 0;
    Plan found{};
    for (int R = 1; R <= kMaxR; ++R) {
        Plan p{};
        p.S = S; p.hl = hl; p.hr = hr; p.R = R; p.Dq = Dq;
        p.n_own = ((S + R - 1) / R + 3) / 4 * 4;
        if (p.n_own > 64 * kMaxBlocks) continue;
        if (R > 1 && ((R - 1) * p.n_own >= S || hl > p.n_own || hr > p.n_own)) break;      // (smaller shares only get worse)
        p.n_jg = p.n_own / 4;
        p.nblk = (p.n_jg + 15) / 16;
        p.waves = 4 * p.nblk;
        p.w_rows = p.n_own + 4 * Dq - 1;
        p.ig_stride = 4 * p.w_rows;
        while (p.ig_stride % 64 != 4) p.ig_stride += 4;
        p.td_off = 0;
        p.w_off = p.td_off + Dq * p.n_own * 16;
        p.m_off = p.w_off + 4 * p.ig_stride * 4;
        p.misc_off = p.m_off + 16 * p.n_own * 4;
        p.lds_bytes = p.misc_off + 256;
        if (p.lds_bytes > kLdsBytes) continue;
        p.rounds = R > 1 ? ((hl + hr) * 8 + 64 * p.waves - 1) / (64 * p.waves) : 0;
        if (p.rounds > kMaxRounds) continue;
        if (best == 0 || R <= want) { best = R; found = p; }
        if (R >= want) break;
    }
    if (best == 0) return false;
    pl = found;
    // the waves' dquads: per 64-next block the own-rows range [qa, qb) and the halo ranges [0, qa), [qb, Dq), four ways
    for (int blk = 0; blk < pl.nblk; ++blk) {
        const int a_lo = 64 * blk, a_hi = std::min(64 * blk + 64, pl.n_own);
        int qa = 0, qb = Dq;
        if (pl.R > 1) {
            qa = std::max(0, (hl - a_lo + 3) / 4);                      // first dquad whose rows start inside the own rows
            const int top = hl + pl.n_own - 3 - a_hi;                   // a_hi + 4 q + 2 < hl + n_own
            qb = top < 0 ? 0 : top / 4 + 1;
            qb = std::min(qb, Dq);
            qa = std::min(qa, qb);
        }
        const int nA = qb - qa, nB = Dq - nA;
        int a_at = qa, b_at = 0;
        for (int p = 0; p < 4; ++p) {
            short *sg = pl.seg[4 * blk + p];
            const int na = nA / 4 + (p < nA % 4 ? 1 : 0);
            const int nb = nB / 4 + (p >= 4 - nB % 4 ? 1 : 0);
            const int a_mid = a_at + (na + 1) / 2;
            sg[0] = (short)a_at; sg[1] = (short)a_mid; sg[2] = (short)a_mid; sg[3] = (short)(a_at + na);
            a_at += na;
            // halo dquads in the order [0, qa) then [qb, Dq): b_at counts through their concatenation
            const int b_end = b_at + nb;
            const int l0 = std::min(b_at, qa), l1 = std::min(b_end, qa);
            const int r0 = std::max(b_at, qa) - qa + qb, r1 = std::max(b_end, qa) - qa + qb;
            sg[4] = (short)l0; sg[5] = (short)l1; sg[6] = (short)r0; sg[7] = (short)r1;
            b_at = b_end;
        }
    }
    return true;
}

__device__ __forceinline__ float4 lds_f4(const char *p) { return *reinterpret_cast<const float4 *>(p); }

template <int K>
__device__ __forceinline__ float comp(const float4 &v) {
    if constexpr (K == 0) return v.x;
    else if constexpr (K == 1) return v.y;
    else if constexpr (K == 2) return v.z;
    else return v.w;
}

// One dquad of one lane: 4 diagonals x 4 next-states x 4 items.  Window rows of the dquad: m = 0 .. 6 (row of diagonal d,
// next-state k: m = d + k) in slots (4 PH + m) & 7, PH = parity of the dquad; rows 0 .. 2 are the previous dquad's 4 .. 6.
template <int PH, int K>
__device__ __forceinline__ void dquad_row(float (&acc)[16], const float4 (&w)[8], const float4 (&t)[4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 r0 = w[(4 * PH + 2 * h + K) & 7], r1 = w[(4 * PH + 2 * h + 1 + K) & 7];
        const float t0 = comp<K>(t[2 * h]), t1 = comp<K>(t[2 * h + 1]);
        acc[4 * K + 0] = fmaxf(fmaxf(acc[4 * K + 0], r0.x + t0), r1.x + t1);
        acc[4 * K + 1] = fmaxf(fmaxf(acc[4 * K + 1], r0.y + t0), r1.y + t1);
        acc[4 * K + 2] = fmaxf(fmaxf(acc[4 * K + 2], r0.z + t0), r1.z + t1);
        acc[4 * K + 3] = fmaxf(fmaxf(acc[4 * K + 3], r0.w + t0), r1.w + t1);
    }
}

template <int PH>
__device__ __forceinline__ void dquad(float (&acc)[16], float4 (&w)[8], const char *tp, const char *wp) {
    float4 t[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) t[d] = lds_f4(tp + 16 * d);
#pragma unroll
    for (int m = 3; m < 7; ++m) w[(4 * PH + m) & 7] = lds_f4(wp + 16 * m);
    dquad_row<PH, 0>(acc, w, t);
    dquad_row<PH, 1>(acc, w, t);
    dquad_row<PH, 2>(acc, w, t);
    dquad_row<PH, 3>(acc, w, t);
}

// dquads [lo, hi) of a lane (wave-uniform bounds): tq0 = Tq[0][jg], w0 = W[ig][4 jg]
__device__ __forceinline__ void scan(float (&acc)[16], float4 (&w)[8], const char *tq0, const char *w0, int lo, int hi,
                                     int tq_step) {
    if (lo >= hi) return;
    const char *tp = tq0 + (size_t)lo * tq_step, *wp = w0 + (size_t)lo * 64;
    int q = lo;
    if (q & 1) {
#pragma unroll
        for (int m = 0; m < 3; ++m) w[(4 + m) & 7] = lds_f4(wp + 16 * m);
        dquad<1>(acc, w, tp, wp);
        ++q;
        tp += tq_step;
        wp += 64;
    } else {
#pragma unroll
        for (int m = 0; m < 3; ++m) w[m] = lds_f4(wp + 16 * m);
    }
    for (; q + 1 < hi; q += 2) {
        dquad<0>(acc, w, tp, wp);
        dquad<1>(acc, w, tp + tq_step, wp + 64);
        tp += 2 * tq_step;
        wp += 128;
    }
    if (q < hi) dquad<0>(acc, w, tp, wp);
}

#ifndef BAND_ABL
#define BAND_ABL 0       // build-time ablations (timing only, results wrong): 1 no waiting for the halo, 2 no history stores
#endif

// grid = tiles of the group x R, block = 64 * pl.waves, dynamic LDS = pl.lds_bytes
__global__ __launch_bounds__(64 * kMaxWaves) void band_forward_kernel(Group grp, Exchange ex, Plan pl,
                                                                      const float *__restrict__ trans,
                                                                      const float *__restrict__ initial) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *const tq = reinterpret_cast<float *>(lds + pl.td_off);
    float *const wq = reinterpret_cast<float *>(lds + pl.w_off);
    float *const mq = reinterpret_cast<float *>(lds + pl.m_off);
    int *const sframes = reinterpret_cast<int *>(lds + pl.misc_off);      // [16] frames per item (0 past the batch)
    int *const sitem = sframes + kNI;                                      // [16] item numbers (a valid one past the batch)
    int *const smisc = sitem + kNI;                                        // [0] ticket, [1] gave up waiting

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = 64 * pl.waves;
    const int S = pl.S, hl = pl.hl, hr = pl.hr, R = pl.R, n_own = pl.n_own, n_jg = pl.n_jg;

    if (tid == 0) {
        smisc[0] = (int)__hip_atomic_fetch_add(ex.control, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        smisc[1] = 0;
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(smisc[0]);
    const int cid = ticket / R, member = ticket - cid * R;
    const int code = grp.tile_map[cid];
    const int bk = code >> 20, tile = code & 0xfffff;
    const Batch &bat = grp.batch[bk];
    const float *__restrict__ obs = bat.obs;
    float *__restrict__ hist = bat.hist;
    const int B = bat.B, T = bat.T;
    const int b0 = tile * kNI;
    const int j0 = member * n_own;

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
    // this member's slab of the band, diagonal-major; everything outside the matrix or the band is -inf
    {
        const int Dd = 4 * pl.Dq;
        for (int e = tid; e < n_own * Dd; e += nthreads) {
            const int a = e / Dd, dd = e - a * Dd;
            const int j = j0 + a, i = j + dd - hl;
            const bool in = j < S && i >= 0 && i < S && dd <= hl + hr;
            tq[((((dd >> 2) * n_jg + (a >> 2)) * 4 + (dd & 3)) << 2) + (a & 3)] = in ? trans[(size_t)j * S + i] : -INFINITY;
        }
        for (int e = tid; e < 4 * pl.ig_stride; e += nthreads) wq[e] = 0.0f;
        for (int e = tid; e < 16 * n_own; e += nthreads) mq[e] = -INFINITY;
    }
    __syncthreads();
    int fmax = 0;
#pragma unroll
    for (int it = 0; it < kNI; ++it) fmax = max(fmax, sframes[it]);

    // ---- the thread as a finisher: one (item group, own row) per thread --------------------------------------------
    const bool fin = tid < 4 * n_own;
    const int fig = fin ? tid / n_own : 0, fa = fin ? tid - fig * n_own : 0;
    const int fj = j0 + fa;
    const bool fin_row = fin && fj < S;
    size_t item_at[4];           // element offset of item bb's row 0, state fj
    int flen[4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
        item_at[bb] = (size_t)sitem[4 * fig + bb] * T * S + (fin_row ? fj : 0);
        flen[bb] = sframes[4 * fig + bb];
    }
    const int fm_at = ((fa & 3) * 4) * n_own + (fa >> 2) * 4 + fig;            // M[(k * 4 + bb)][jg * 4 + ig], bb = 0
    float *const fw_at = wq + fig * pl.ig_stride + 4 * (hl + fa);
    char *const xtile = ex.xchg[bk] + (size_t)tile * xchg_tile_bytes(S);
    const unsigned xpar = (unsigned)(4 * S * 32);                                // bytes of one parity
    const __amdgpu_buffer_rsrc_t xbuf = buffer_of(xtile, 2u * xpar);
    const int fx_at = (fig * S + fj) * 32;
    const bool publishes = fin_row && R > 1;
    float ob[4] = {0.f, 0.f, 0.f, 0.f};

    auto finish = [&](int t, const float (&best)[4], bool more) {
        // post'[j] = obs[t][j] + max (viterbi.cpp:102) -> the window, the history, the neighbours
        float v[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) v[bb] = ob[bb] + best[bb];
        *reinterpret_cast<float4 *>(fw_at) = make_float4(v[0], v[1], v[2], v[3]);
        if (fin_row && !(BAND_ABL & 2)) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
                if (t < flen[bb]) hist[item_at[bb] + (size_t)t * S] = v[bb];
        }
        if (publishes && more) {
            const unsigned tag = (unsigned)t + 1u;
            const int at = (int)((unsigned)(t & 1) * xpar) + fx_at;
            v4u g0 = {__float_as_uint(v[0]), tag, __float_as_uint(v[1]), tag};
            v4u g1 = {__float_as_uint(v[2]), tag, __float_as_uint(v[3]), tag};
            __builtin_amdgcn_raw_buffer_store_b128(g0, xbuf, at, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(g1, xbuf, at + 16, 0, 16);
        }
        if (more && fin_row) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) ob[bb] = obs[item_at[bb] + (size_t)(t + 1) * S];
        }
    };

    // t = 0: posterior row 0 = obs[b][0][:] + initial (viterbi.cpp:72-76)
    if (fin) {
        float first[4];
        const float ini = fin_row ? initial[fj] : 0.0f;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            ob[bb] = fin_row ? obs[item_at[bb]] : 0.0f;
            first[bb] = ini;
        }
        finish(0, first, fmax > 1);
    }

    // ---- the thread as a reader of halo granules --------------------------------------------------------------------
    int hx_at[kMaxRounds], hw_at[kMaxRounds];           // byte offset in a parity of the exchange (-1: none), LDS byte address
#pragma unroll
    for (int r = 0; r < kMaxRounds; ++r) {
        const int p = tid + r * nthreads;
        const int half = p & 1, ig = (p >> 1) & 3, hrow = p >> 3;
        const int wrow = hrow < hl ? hrow : hrow + n_own;
        const int grow = j0 - hl + wrow;
        const bool valid = r < pl.rounds && hrow < hl + hr && grow >= 0 && grow < S;
        hx_at[r] = valid ? ((ig * S + grow) * 2 + half) * 16 : -1;
        hw_at[r] = pl.w_off + (ig * pl.ig_stride + 4 * wrow + 2 * half) * 4;
    }

    // ---- the thread as a lane of the scan: 4 next-states (jg) x 4 items (ig) ---------------------------------------------
    const int ig = lane & 3, jgl = lane >> 2;
    const int blk = wave >> 2;
    const int jg_raw = 16 * blk + jgl;
    const bool scans = jg_raw < n_jg;
    const int jg = scans ? jg_raw : n_jg - 1;
    const char *const tq0 = lds + pl.td_off + jg * 64;
    const char *const w0 = lds + pl.w_off + (ig * pl.ig_stride + 16 * jg) * 4;
    const int tq_step = n_jg * 64;
    float *const m_at = mq + jg * 4 + ig;
    const short *const sg = pl.seg[wave];
    const int s0 = sg[0], s1 = sg[1], s2 = sg[2], s3 = sg[3], s4 = sg[4], s5 = sg[5], s6 = sg[6], s7 = sg[7];
    bool gave_up = false;

    __syncthreads();
    for (int t = 1; t < fmax; ++t) {
        float acc[16];
        float4 w[8];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = -INFINITY;
        scan(acc, w, tq0, w0, s0, s1, tq_step);
        // the neighbours' rows t - 1: asked for now, looked at behind the second run of own-row dquads
        v4u got[kMaxRounds];
        const int par_at = (int)((unsigned)((t - 1) & 1) * xpar);
        if (R > 1) {
#pragma unroll
            for (int r = 0; r < kMaxRounds; ++r)
                if (hx_at[r] >= 0) got[r] = __builtin_amdgcn_raw_buffer_load_b128(xbuf, par_at + hx_at[r], 0, 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        scan(acc, w, tq0, w0, s2, s3, tq_step);
        if (R > 1) {
            const unsigned tag = (unsigned)t;
            unsigned long long since = 0ull;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int r = 0; r < kMaxRounds; ++r)
                    if (hx_at[r] >= 0) ok = ok && got[r].y == tag && got[r].w == tag;
                if (__all(ok) || gave_up || (BAND_ABL & 1)) break;
                const unsigned long long now = wall_clock64();
                if (since == 0ull) since = now;
                if (now - since >= ex.wait_ticks) { gave_up = true; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int r = 0; r < kMaxRounds; ++r)
                    if (hx_at[r] >= 0 && !(got[r].y == tag && got[r].w == tag))
                        got[r] = __builtin_amdgcn_raw_buffer_load_b128(xbuf, par_at + hx_at[r], 0, 16);
            }
#pragma unroll
            for (int r = 0; r < kMaxRounds; ++r)
                if (hx_at[r] >= 0)
                    *reinterpret_cast<float2 *>(lds + hw_at[r]) = make_float2(__uint_as_float(got[r].x), __uint_as_float(got[r].z));
            __syncthreads();            // the halo rows t - 1 are in the window
            scan(acc, w, tq0, w0, s4, s5, tq_step);
            scan(acc, w, tq0, w0, s6, s7, tq_step);
        }
        if (scans) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                __builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float *)(m_at + e * n_own), acc[e], 0, 0, false);
        }
        __syncthreads();                // every wave is done with the window; M holds the maxima
        if (fin) {
            float best[4];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                best[bb] = mq[fm_at + bb * n_own];
                mq[fm_at + bb * n_own] = -INFINITY;
            }
            finish(t, best, t + 1 < fmax);
        }
        __syncthreads();                // the window holds the own rows t
    }
    if (gave_up && lane == 0) {
        ex.failed[cid] = 1u;
        atomicAdd(&grp.stats[127], 1u);
    }
}

// The safety net behind a band launch: a tile whose members gave up waiting is decoded again by ONE workgroup, four
// items at a time, posterior rows ping-pong in the LDS, no hand-offs.  grid = tiles, block = 1024, LDS = 32 S bytes.
__global__ __launch_bounds__(1024) void band_repair_kernel(Group grp, const unsigned *__restrict__ failed,
                                                           const float *__restrict__ trans, const float *__restrict__ initial,
                                                           int S, int hl, int hr) {
    if (failed[blockIdx.x] == 0u) return;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float4 *rows = reinterpret_cast<float4 *>(lds);
    __shared__ int sframes[kNI], sitem[kNI];
    const int tid = threadIdx.x;
    const int code = grp.tile_map[blockIdx.x];
    const Batch &bat = grp.batch[code >> 20];
    const int B = bat.B, T = bat.T, b0 = (code & 0xfffff) * kNI;
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
    __syncthreads();
    for (int g = 0; g < 4; ++g) {
        size_t at[4];
        int len[4], longest = 0;
        for (int bb = 0; bb < 4; ++bb) {
            at[bb] = (size_t)sitem[4 * g + bb] * T * S;
            len[bb] = sframes[4 * g + bb];
            longest = max(longest, len[bb]);
        }
        __syncthreads();
        for (int j = tid; j < S; j += 1024) {
            float v[4];
            for (int bb = 0; bb < 4; ++bb) {
                v[bb] = bat.obs[at[bb] + j] + initial[j];
                if (len[bb] > 0) bat.hist[at[bb] + j] = v[bb];
            }
            rows[j] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        for (int t = 1; t < longest; ++t) {
            const float4 *cur = rows + (size_t)((t - 1) & 1) * S;
            float4 *nxt = rows + (size_t)(t & 1) * S;
            for (int j = tid; j < S; j += 1024) {
                float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                const int lo = max(0, j - hl), hi = min(S - 1, j + hr);
                const float *row = trans + (size_t)j * S;
                for (int i = lo; i <= hi; ++i) {
                    const float tv = row[i];
                    const float4 p = cur[i];
                    best[0] = fmaxf(best[0], p.x + tv);
                    best[1] = fmaxf(best[1], p.y + tv);
                    best[2] = fmaxf(best[2], p.z + tv);
                    best[3] = fmaxf(best[3], p.w + tv);
                }
                float v[4];
                for (int bb = 0; bb < 4; ++bb) {
                    v[bb] = bat.obs[at[bb] + (size_t)t * S + j] + best[bb];
                    if (t < len[bb]) bat.hist[at[bb] + (size_t)t * S + j] = v[bb];
                }
                nxt[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
            __syncthreads();
        }
    }
}

// zero the exchange buffers of a launch (tags of an earlier decode must not be taken for this one's) and its control words
struct ClearJobs {
    char *xchg[resident::kMaxBatches];
    size_t bytes[resident::kMaxBatches];
    unsigned *words;
    int nwords;
    int n;
};
__global__ __launch_bounds__(256) void clear_exchange_kernel(ClearJobs jobs) {
    const int k = blockIdx.y;
    if (k < jobs.n) {
        uint4 *p = reinterpret_cast<uint4 *>(jobs.xchg[k]);
        const size_t n = jobs.bytes[k] / 16;
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) p[e] = zero;
    }
    if (k == 0)
        for (int e = blockIdx.x * 256 + threadIdx.x; e < jobs.nwords; e += gridDim.x * 256) jobs.words[e] = 0u;
}

// ---------------------------------------------------------------------------------------
// final state, tail fill and backtrace of every item of the group: the first argmax of fl(hist[t-1][i] + trans[j][i]) over
// the band of the state on the path (viterbi.cpp:81-100, 140-160), as lazy::backtrace_ranged_kernel finds it inside a
// row's finite range.  grid = items of the group, one wave per item; S % 4 == 0, S <= 256 NQ, hl + hr + 4 <= 512.
// ---------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(64) void group_backtrace_band_kernel(Group grp, const float *__restrict__ trans, int S, int hl,
                                                                  int hr) {
    const Batch &bat = grp.batch[resident::batch_of_item(grp, blockIdx.x)];
    const int b = (int)blockIdx.x - bat.item0, lane = threadIdx.x, T = bat.T;
    const float *h = bat.hist + (size_t)b * T * S;
    int32_t *o = bat.out + (size_t)b * T;
    int f = bat.frames[b];
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
        j = lazy::wave_first_argmax4<NQ>(last, lane, S);       // final state = first argmax of the last row (viterbi.cpp:218)
    }
    for (int tt = f - 1 + lane; tt < T; tt += 64) o[tt] = j;    // viterbi.cpp:219-221
    const float4 none = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int tt = f - 1; tt >= 1; --tt) {
        const int lo4 = max(0, j - hl) & ~3, hi = min(S, j + hr + 1);
        const float *tr = trans + (size_t)j * S, *hrow = h + (size_t)(tt - 1) * S;
        float4 cand[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = lo4 + 4 * lane + 256 * q;
            if (i < hi) {
                const float4 t4 = *reinterpret_cast<const float4 *>(tr + i);
                const float4 p4 = *reinterpret_cast<const float4 *>(hrow + i);
                cand[q] = make_float4(p4.x + t4.x, p4.y + t4.y, p4.z + t4.z, p4.w + t4.w);
            } else {
                cand[q] = none;
            }
        }
        const float m = wavered::wave_reduce_f32(
            __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(cand[0].x, cand[0].y), __builtin_fmaxf(cand[0].z, cand[0].w)),
                            __builtin_fmaxf(__builtin_fmaxf(cand[1].x, cand[1].y), __builtin_fmaxf(cand[1].z, cand[1].w))),
            wavered::MaxOp());
        int k = lazy::kSentinel;
#pragma unroll
        for (int q = 1; q >= 0; --q) {
            const int i = lo4 + 4 * lane + 256 * q;
            int kq = cand[q].w == m ? i + 3 : lazy::kSentinel;
            kq = cand[q].z == m ? i + 2 : kq;
            kq = cand[q].y == m ? i + 1 : kq;
            kq = cand[q].x == m ? i : kq;
            k = min(k, kq);
        }
        k = wavered::wave_min_i32(k);
        j = m == -INFINITY ? 0 : k;          // (every candidate -inf: the reference's scan keeps prev-state 0)
        if (lane == 0) o[tt - 1] = j;
    }
}

// reach of a matrix: *left = max over finite entries of (j - i), *right = max of (i - j), both >= 0; grid = S, block = 64;
// `reach` zeroed by the caller
__global__ __launch_bounds__(64) void band_reach_kernel(const float *__restrict__ trans, int32_t *__restrict__ reach, int S) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const float *row = trans + (size_t)j * S;
    int left = 0, right = 0;
    for (int i = lane; i < S; i += 64)
        if (row[i] != -INFINITY) {
            left = max(left, j - i);
            right = max(right, i - j);
        }
    left = -wavered::wave_min_i32(-left);
    right = -wavered::wave_min_i32(-right);
    if (lane == 0) {
        atomicMax(reach, left);
        atomicMax(reach + 1, right);
    }
}

}  // namespace band
