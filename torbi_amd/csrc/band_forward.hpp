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
// 16-byte granules {v0, tag, v1, tag} (tag = timestep + 1; the granule's two 8-byte halves are each {value, tag}:
// MI355X_MICROARCH.md "R2's granule") through a per-tile exchange buffer by timestep parity -- one trip, no flag, no
// fence.  The members of a tile are drawn from ONE dispatch class (workgroups b, b + 8, b + 16, ... -- the GPU places
// them on one XCD) and say at kernel entry which XCD they run on: when all R agree the granules are PLAIN stores that stay
// in that XCD's L2 and L1-bypassing loads that hit it (the exchange buffer is rewritten every other timestep and never
// leaves the L2); when they do not -- nothing promises the placement -- write-through stores carry them across XCDs.
// Write-through granules cost 2.2 of 9 us per timestep at 512 x 1440 (23 KB per workgroup and timestep, 5.9 MB across
// the chip, every byte through the fabric; profiles/r05_band_ablations.txt).  Each wave evaluates the dquads that touch only its member's own rows FIRST (50-66 %
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
    float background;        // every entry outside the band: -inf, or ONE constant (band_forward_kernel<true>; the head of
                             // band_tile_forward.hpp has the rule and its proof)
    // per wave, in dquads.  Before the halo is in the window: [0],[1) and [2],[3) read only the member's own rows (the halo
    // granules are asked for between the two runs) -- the same number of dquads for every wave, so that no wave idles at
    // the barrier in the middle of the timestep; behind it: [4],[5) the wave's other own-row dquads, [6],[7) and [8],[9)
    // the ones that read halo rows
    short seg[kMaxWaves][10];
};

struct Exchange {
    char *xchg[resident::kMaxBatches];   // per batch: [tiles] x { [2 parities][16 items][2 halves][S / 4] granules of 16 bytes,
                                         // [16] the members' XCDs + 1 (256 bytes) }
    int tile0, tiles;                    // this launch decodes tiles [tile0, tiles) of the group's tile map (tile0 % 8 == 0)
    unsigned *control;                   // [0 .. 7] tickets drawn per dispatch class IN THIS LAUNCH (zeroed before it)
    unsigned *failed;                    // [tiles of the group] set by a member that gave up waiting (zeroed before the launch)
    unsigned long long wait_ticks;       // budget of one wait (100 MHz ticks)
};

// (behind the granule planes and the members' XCDs: [2 parities][kMaxR members][16 items] 16-byte records {largest own posterior
// of the row as an ordered integer, lowest state attaining it, highest, timestep} -- band_forward_kernel<true>)
constexpr size_t kStatBytes = (size_t)2 * 16 * kNI * 16;
__host__ __device__ inline size_t xchg_tile_bytes(int S) { return (size_t)2 * kNI * (size_t)S * 8 + 256 + kStatBytes; }
__host__ __device__ inline size_t xchg_bytes(int B, int S) { return (size_t)((B + kNI - 1) / kNI) * xchg_tile_bytes(S); }

// The plan for `tiles` tiles on `cus` compute units, or false when the band kernel does not cover the shape.  A dispatch
// class (workgroups b, b + 8, ...) runs on ONE XCD and holds R x ceil(tiles / 8) members, all of which must be resident at
// once (they wait for each other inside the launch): the plan takes the largest R that fits the XCD's cus / 8 units; when
// even the smallest R the LDS allows does not, the caller decodes tiles_per_launch(pl, cus) tiles per launch.
inline int tiles_per_launch(const Plan &pl, int cus) { return pl.R == 1 ? 1 << 30 : 8 * ((cus / 8) / pl.R); }
inline bool make_plan(int S, int hl, int hr, int tiles, int cus, Plan &pl, float background = -INFINITY) {
    if (S < 64 || S % 4 != 0 || hl < 0 || hr < 0 || hl >= S || hr >= S || tiles < 1) return false;
    if (hl + hr + 4 > kMaxWindow) return false;
    const int Dq = (hl + hr + 1 + 3) / 4;
    const int want = std::max(1, std::min(kMaxR, (cus / 8) / ((tiles + 7) / 8)));
    int best = 0;
    Plan found{};
    for (int R = 1; R <= kMaxR; ++R) {
        Plan p{};
        p.S = S; p.hl = hl; p.hr = hr; p.R = R; p.Dq = Dq;
        p.background = background;
        if (background != -INFINITY && R == 1) continue;          // (one member per tile: the whole-tile kernel's case)
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
        p.lds_bytes = p.misc_off + 256 + (background != -INFINITY ? 512 : 0);      // (+ four 16 x 8-byte key tables)
        if (p.lds_bytes > kLdsBytes) continue;
        p.rounds = R > 1 ? (((hl + 1) / 2 + (hr + 1) / 2) * kNI + 64 * p.waves - 1) / (64 * p.waves) : 0;
        if (p.rounds > kMaxRounds) continue;
        if (best == 0 || R <= want) { best = R; found = p; }
        if (R >= want) break;
    }
    if (best == 0) return false;
    pl = found;
    // the waves' dquads: per 64-next block the own-rows range [qa, qb) and the halo ranges [0, qa), [qb, Dq), four ways
    int own[kMaxWaves][2], pre = Dq;
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
            own[4 * blk + p][0] = a_at;
            own[4 * blk + p][1] = a_at + na;
            pre = std::min(pre, na);
            a_at += na;
            // halo dquads in the order [0, qa) then [qb, Dq): b_at counts through their concatenation
            const int b_end = b_at + nb;
            sg[6] = (short)std::min(b_at, qa); sg[7] = (short)std::min(b_end, qa);
            sg[8] = (short)(std::max(b_at, qa) - qa + qb); sg[9] = (short)(std::max(b_end, qa) - qa + qb);
            b_at = b_end;
        }
    }
    if (pl.R == 1) pre = Dq;          // (no halo: no barrier in the middle)
    for (int wv = 0; wv < pl.waves; ++wv) {
        short *sg = pl.seg[wv];
        const int lo = own[wv][0], hi = own[wv][1], cut = std::min(hi, lo + pre), mid = lo + (cut - lo + 1) / 2;
        sg[0] = (short)lo; sg[1] = (short)mid; sg[2] = (short)mid; sg[3] = (short)cut; sg[4] = (short)cut; sg[5] = (short)hi;
    }
    return true;
}

#ifndef BAND_ABL
#define BAND_ABL 0       // build-time ablations (timing only, results wrong): 1 no waiting for the halo, 2 no history stores,
                         // 4 no merge through M, 8 no exchange stores, 16 the scans alone (no hand-off, merge, finish, barriers),
                         // 32 write-through granules even inside one XCD (results right), 64 scans without LDS reads, 128 scans without arithmetic,
                         // 256 / 512 without the barrier in the middle of the timestep / behind the merge
#endif

__device__ __forceinline__ float4 lds_f4(const char *p) {
    if (BAND_ABL & 64) return make_float4(1.f, 2.f, 3.f, 4.f);
    return *reinterpret_cast<const float4 *>(p);
}

template <int K>
__device__ __forceinline__ float comp(const float4 &v) {
    if constexpr (K == 0) return v.x;
    else if constexpr (K == 1) return v.y;
    else if constexpr (K == 2) return v.z;
    else return v.w;
}

// One dquad of one lane: 4 diagonals x 4 next-states x 4 items.  Window rows of the dquad: m = 0 .. 6 (row of diagonal d,
// next-state k: m = d + k) in slots (4 PH + m) & 7, PH = parity of the dquad; rows 0 .. 2 are the previous dquad's 4 .. 6.
//
// Software pipeline without a register to spare: a dquad's operands are all ON THEIR WAY when it starts, and while it
// runs it asks for the NEXT dquad's -- each into a register that has just seen its last use:
//     start:                      next row 3 -> the one free slot of the ring
//     diagonals 0, 1 (rows 0 .. 4): then next t[0], t[1] and next rows 4, 5 -> the slots of rows 0, 1
//     diagonals 2, 3 (rows 2 .. 6): then next t[2], t[3] and next row 6 -> the slot of row 2
// so every ds_read has at least half a dquad (128 issue cycles) to land, and a wave that has its SIMD to itself -- the
// last of three to finish a phase -- still runs at the vector ALU's rate.
template <int PH, int D0, int K>
__device__ __forceinline__ void dquad_cells(float (&acc)[16], const float4 (&w)[8], const float4 (&t)[4]) {
    const float4 r0 = w[(4 * PH + D0 + K) & 7], r1 = w[(4 * PH + D0 + 1 + K) & 7];
    const float t0 = comp<K>(t[D0]), t1 = comp<K>(t[D0 + 1]);
    if (BAND_ABL & 128) {
        if (D0 == 2) asm volatile("" :: "v"(r0.x), "v"(r1.w), "v"(t0), "v"(t1));
        return;
    }
    acc[4 * K + 0] = fmaxf(fmaxf(acc[4 * K + 0], r0.x + t0), r1.x + t1);
    acc[4 * K + 1] = fmaxf(fmaxf(acc[4 * K + 1], r0.y + t0), r1.y + t1);
    acc[4 * K + 2] = fmaxf(fmaxf(acc[4 * K + 2], r0.z + t0), r1.z + t1);
    acc[4 * K + 3] = fmaxf(fmaxf(acc[4 * K + 3], r0.w + t0), r1.w + t1);
}

// tp / wp: this dquad's Tq block and first window row; tn / wn: the next dquad's (anything readable behind the last one)
template <int PH>
__device__ __forceinline__ void dquad(float (&acc)[16], float4 (&w)[8], float4 (&t)[4], const char *tn, const char *wn) {
    // (the requests of a half are pinned AHEAD of its arithmetic: left to itself the scheduler sinks them towards their
    // first use, half a dquad later at best)
    w[(4 * PH + 7) & 7] = lds_f4(wn + 16 * 3);
    __builtin_amdgcn_sched_barrier(0);
    dquad_cells<PH, 0, 0>(acc, w, t);
    dquad_cells<PH, 0, 1>(acc, w, t);
    dquad_cells<PH, 0, 2>(acc, w, t);
    dquad_cells<PH, 0, 3>(acc, w, t);
    __builtin_amdgcn_sched_barrier(0);
    t[0] = lds_f4(tn);
    t[1] = lds_f4(tn + 16);
    w[(4 * PH + 8) & 7] = lds_f4(wn + 16 * 4);
    w[(4 * PH + 9) & 7] = lds_f4(wn + 16 * 5);
    __builtin_amdgcn_sched_barrier(0);
    dquad_cells<PH, 2, 0>(acc, w, t);
    dquad_cells<PH, 2, 1>(acc, w, t);
    dquad_cells<PH, 2, 2>(acc, w, t);
    dquad_cells<PH, 2, 3>(acc, w, t);
    __builtin_amdgcn_sched_barrier(0);
    t[2] = lds_f4(tn + 32);
    t[3] = lds_f4(tn + 48);
    w[(4 * PH + 10) & 7] = lds_f4(wn + 16 * 6);
    __builtin_amdgcn_sched_barrier(0);
}

// dquads [lo, hi) of a lane (wave-uniform bounds): tq0 = Tq[0][jg], w0 = W[ig][4 jg]; `hook` runs ahead of dquad `at`
template <typename Hook>
__device__ __forceinline__ void scan(float (&acc)[16], float4 (&w)[8], const char *tq0, const char *w0, int lo, int hi,
                                     int tq_step, int at, Hook hook) {
    if (lo >= hi) { if (at >= lo) hook(); return; }
    const char *tp = tq0 + (size_t)lo * tq_step, *wp = w0 + (size_t)lo * 64;
    float4 t[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) t[d] = lds_f4(tp + 16 * d);
    int q = lo;
    if (q & 1) {
#pragma unroll
        for (int m = 0; m < 7; ++m) w[(4 + m) & 7] = lds_f4(wp + 16 * m);
        if (q == at) hook();
        dquad<1>(acc, w, t, tp + tq_step, wp + 64);
        ++q;
        tp += tq_step;
        wp += 64;
    } else {
#pragma unroll
        for (int m = 0; m < 7; ++m) w[m] = lds_f4(wp + 16 * m);
    }
    for (; q + 1 < hi; q += 2) {
        if (q == at) hook();
        dquad<0>(acc, w, t, tp + tq_step, wp + 64);
        if (q + 1 == at) hook();
        dquad<1>(acc, w, t, tp + 2 * tq_step, wp + 128);
        tp += 2 * tq_step;
        wp += 128;
    }
    if (q < hi) {
        if (q == at) hook();
        dquad<0>(acc, w, t, tp + tq_step, wp + 64);
        ++q;
    }
    if (at >= q) hook();         // (a hook at or behind the end of the run)
}
__device__ __forceinline__ void scan(float (&acc)[16], float4 (&w)[8], const char *tq0, const char *w0, int lo, int hi,
                                     int tq_step) {
    scan(acc, w, tq0, w0, lo, hi, tq_step, -1, [] {});
}

#ifdef BAND_STAMP
// build-time instrumentation (tools/band_stamps.py): per-wave cycle sums of the phases of a timestep
constexpr int kPhases = 12;
__device__ unsigned long long g_phase[1024 * kMaxWaves * kPhases];
#define BSTAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); bacc[i] += now_ - blast; blast = now_; }
#else
#define BSTAMP(i)
#endif

// monotone map float -> unsigned (and back): larger value, larger key; NaNs are the alarm's business (nonfinite.hpp)
__device__ __forceinline__ unsigned ordered(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unordered(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o); }

// grid = tiles of the group x R, block = 64 * pl.waves, dynamic LDS = pl.lds_bytes
// BG: ONE constant outside the band instead of -inf (band_tile_forward.hpp, head).  A member knows its own rows only: it
// leaves the largest posterior of every row it finishes and the lowest / highest state attaining it (64-bit keys, one LDS
// atomic maximum each: {ordered value, ~state} and {ordered value, state}) as one 16-byte record per item in the exchange
// when the next timestep opens; every member takes all R records of an item in just before the barrier that closes its
// scans (they were written a timestep's scans ago: the first look finds them), combines them with the same keys, and its
// finishers decide their outputs as the whole-tile kernel does.  No barrier more than the -inf instance has.
template <bool BG = false>
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
    // BG: own rows' keys of the row being finished; all members' keys of the row before (lowest / highest state in the low word)
    unsigned long long *const lkey_lo = reinterpret_cast<unsigned long long *>(lds + pl.misc_off + 256);
    unsigned long long *const lkey_hi = lkey_lo + kNI, *const gkey_lo = lkey_lo + 2 * kNI, *const gkey_hi = lkey_lo + 3 * kNI;
    const float cbg = pl.background;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = 64 * pl.waves;
    const unsigned long long clock_0 = clock64(), wall_0 = wall_clock64();      // (stats[120], [121]: the clock under this load)
    const int S = pl.S, hl = pl.hl, hr = pl.hr, R = pl.R, n_own = pl.n_own, n_jg = pl.n_jg;

    // Membership by arrival WITHIN a dispatch class (blockIdx mod 8): tile = class + 8 x (ticket / R), member = ticket mod R.
    // A class holds exactly R x ceil(tiles / 8) workgroups (grid = 8 x that), so every tile gets its R members whatever the
    // dispatch order; the GPU places a class on one XCD (observed, not promised: the members compare notes below).
    int cid = blockIdx.x, member = 0;
    if (R > 1) {
        const int cls = blockIdx.x & 7;
        if (tid == 0) {
            smisc[0] = (int)__hip_atomic_fetch_add(ex.control + cls, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            smisc[1] = 0;
        }
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(smisc[0]);
        member = ticket % R;
        cid = ex.tile0 + (ticket / R) * 8 + cls;
        if (cid >= ex.tiles) return;            // (the grid is padded to whole classes)
    }
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
        if (BG) lkey_lo[tid] = lkey_hi[tid] = gkey_lo[tid] = gkey_hi[tid] = 0ull;
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

    // ---- the thread as a finisher: one (item, four consecutive own rows) per thread -----------------------------------
    const bool fin = tid < kNI * n_jg;
    const int fit = fin ? tid / n_jg : 0, fjg = fin ? tid - fit * n_jg : 0;       // tile item, group of four next-states
    const int fig = fit >> 2, fbb = fit & 3;
    const int fj = j0 + 4 * fjg;
    const bool fin_row = fin && fj < S;                  // (S % 4 == 0: the four rows are inside the matrix or none is)
    const size_t item_at = (size_t)sitem[fit] * T * S + (fin_row ? fj : 0);        // element offset of row 0, state fj
    const int flen = sframes[fit];
    float *const fm_at = mq + fbb * n_own + fjg * 4 + fig;                  // M[(k * 4 + bb)][jg * 4 + ig], k = 0
    float *const fw_at = wq + fig * pl.ig_stride + 4 * (hl + 4 * fjg) + fbb;
    char *const xtile = ex.xchg[bk] + (size_t)tile * xchg_tile_bytes(S);
    const unsigned xpar = (unsigned)(kNI * S * 8);               // bytes of one parity: [16 items][2 halves][S / 4] granules
    const __amdgpu_buffer_rsrc_t xbuf = buffer_of(xtile, 2u * xpar);
    const __amdgpu_buffer_rsrc_t sbuf = buffer_of(xtile + 2u * xpar + 256u, (unsigned)kStatBytes);      // (BG: the members' records)
    const int xhalf = (S / 4) * 16;                              // bytes of one half plane of an item
    const int fx_at = fit * 2 * xhalf + (fj >> 2) * 16;          // granule {fj, fj + 1}; {fj + 2, fj + 3} one plane on
    const bool publishes = fin_row && R > 1;
    // do the R members run on one XCD?  Each says where it is (write-through), all read all R answers (bounded wait)
    bool local = false;
    if (R > 1) {
        unsigned *const where = reinterpret_cast<unsigned *>(xtile + 2u * xpar);
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;      // HW_REG_XCC_ID
        if (wave == 0) {
            if (lane == 0) __hip_atomic_store(where + member, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long since = 0ull;
            unsigned seen = xcc + 1u;
            for (;;) {
                if (lane < R) seen = __hip_atomic_load(where + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(seen != 0u)) break;
                const unsigned long long now = wall_clock64();
                if (since == 0ull) since = now;
                if (now - since >= ex.wait_ticks) { if (lane == 0) smisc[1] = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            const bool same = __all(seen == xcc + 1u);
            if (lane == 0) smisc[2] = same ? 1 : 0;
            // (a budget of zero means "do not wait at all": the member reports that it gave up whether or not a poll
            // failed -- the tests of the repair launch rely on it)
            if (lane == 0 && ex.wait_ticks == 0ull) smisc[1] = 1;
        }
        __syncthreads();
        local = smisc[2] != 0 && !(BAND_ABL & 32);
    }
    const bool incomplete = R > 1 && smisc[1] != 0;        // (a member never showed up in time: no waiting for it later either)
    // (the observations of row t + 2 are asked for when row t is finished: a wave's loads return in order, and a halo
    // granule asked for behind a first-touch HBM read would wait for it)
    float4 ob = make_float4(0.f, 0.f, 0.f, 0.f), ob_next = ob;

    // post'[j] = obs[t][j] + max (viterbi.cpp:102).  `settle`: into the window; `send`: to the neighbours and the history
    // (measured: sending behind the barrier that opens the next timestep, so that the stores issue under its first dquads,
    // is 2 % slower -- the granules leave later); `fetch`: the observations of row t + 2 (behind the halo granules of the timestep: a wave's loads return in
    // order, and a granule asked for behind a first-touch HBM read would wait for it)
    bool odd = false;                  // a NaN / +inf posterior value was produced (nonfinite.hpp)
    auto settle = [&](const float4 &best) {
        const float4 v = make_float4(ob.x + best.x, ob.y + best.y, ob.z + best.z, ob.w + best.w);
        odd = odd || nonfinite::odd4(v);
        fw_at[0] = v.x;
        fw_at[4] = v.y;
        fw_at[8] = v.z;
        fw_at[12] = v.w;
        if (BG && fin_row) {            // the largest of the four and where: lowest state in one key, highest in the other
            const float vs[4] = {v.x, v.y, v.z, v.w};
            int fit_ = fit, fj_ = fj;
            asm volatile("" : "+v"(fit_), "+v"(fj_));
            unsigned long long lo = 0ull, hi = 0ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long o = (unsigned long long)ordered(vs[k]) << 32;
                lo = max(lo, o | (unsigned long long)(0xffffffffu - (unsigned)(fj_ + k)));
                hi = max(hi, o | (unsigned long long)(unsigned)(fj_ + k));
            }
            atomicMax(lkey_lo + fit_, lo);
            atomicMax(lkey_hi + fit_, hi);
        }
        return v;
    };
    auto send = [&](int t, const float4 &v, bool more) {
        if (publishes && more && !(BAND_ABL & 8)) {
            const unsigned tag = (unsigned)t + 1u;
            const int at = (int)((unsigned)(t & 1) * xpar) + fx_at;
            v4u g0 = {__float_as_uint(v.x), tag, __float_as_uint(v.y), tag};
            v4u g1 = {__float_as_uint(v.z), tag, __float_as_uint(v.w), tag};
            if (local) {            // one XCD: the granules stay in its L2
                __builtin_amdgcn_raw_buffer_store_b128(g0, xbuf, at, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(g1, xbuf, at + xhalf, 0, 0);
            } else {                // write-through: visible to every XCD
                __builtin_amdgcn_raw_buffer_store_b128(g0, xbuf, at, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(g1, xbuf, at + xhalf, 0, 16);
            }
        }
        if (fin_row && t < flen && !(BAND_ABL & 2)) *reinterpret_cast<float4 *>(hist + item_at + (size_t)t * S) = v;
    };
    auto fetch = [&](int t) {       // called in timestep t: row t is already in `ob_next`
        ob = ob_next;
        if (fin_row && t + 1 < fmax) ob_next = *reinterpret_cast<const float4 *>(obs + item_at + (size_t)(t + 1) * S);
    };

    // t = 0: posterior row 0 = obs[b][0][:] + initial (viterbi.cpp:72-76)
    if (fin) {
        float4 first = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fin_row) {
            ob = *reinterpret_cast<const float4 *>(obs + item_at);
            if (fmax > 1) ob_next = *reinterpret_cast<const float4 *>(obs + item_at + (size_t)S);
            first = make_float4(initial[fj], initial[fj + 1], initial[fj + 2], initial[fj + 3]);
        }
        send(0, settle(first), fmax > 1);
        ob = ob_next;                 // row 1; row 2 is asked for in timestep 1
    }

    // ---- the thread as a reader of halo granules: two consecutive states of one item, {v, tag, v', tag} ---------------------
    int hx_at[kMaxRounds], hw_at[kMaxRounds];           // byte offset in a parity of the exchange (-1: none), LDS byte address
    bool need0[kMaxRounds], need1[kMaxRounds];          // which of the two states the window wants
    {
        const int left_first = (j0 - hl) >> 1;                      // granule g holds states 2 g, 2 g + 1 (j0 is even)
        const int nleft = (j0 >> 1) - left_first;                   // granules left of the own rows
        const int right_first = (j0 + n_own) >> 1, nright = (hr + 1) / 2;
        const int ncg = nleft + nright;                             // granules per item
#pragma unroll
        for (int r = 0; r < kMaxRounds; ++r) {
            const int p = tid + r * nthreads;
            const int item = (p / ncg) & (kNI - 1), c = p % ncg;
            const int g = c < nleft ? left_first + c : right_first + (c - nleft);
            const int st = 2 * g;                                  // first state of the granule
            const int wrow = st - (j0 - hl);                       // its row of the window (may be -1)
            auto wanted = [&](int state) {
                const bool in_halo = (state >= j0 - hl && state < j0) || (state >= j0 + n_own && state < j0 + n_own + hr);
                return in_halo && state >= 0 && state < S;
            };
            const bool listed = r < pl.rounds && p < kNI * ncg;
            need0[r] = listed && wanted(st);
            need1[r] = listed && wanted(st + 1);
            hx_at[r] = (need0[r] || need1[r]) ? (item * 2 + (g & 1)) * xhalf + (g >> 1) * 16 : -1;
            hw_at[r] = pl.w_off + ((item >> 2) * pl.ig_stride + 4 * wrow + (item & 3)) * 4;
        }
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
    const int s0 = sg[0], s1 = sg[1], s2 = sg[2], s3 = sg[3], s4 = sg[4], s5 = sg[5], s6 = sg[6], s7 = sg[7], s8 = sg[8],
              s9 = sg[9];
    bool gave_up = incomplete;

    __syncthreads();
#ifdef BAND_STAMP
    unsigned long long bacc[kPhases] = {};
    unsigned long long blast = __builtin_readcyclecounter();
#endif
    bool undecided = false;            // BG: an output neither test of band_tile_forward.hpp's head decided
    for (int t = 1; t < fmax; ++t) {
        float acc[16];
        float4 w[8];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = -INFINITY;
        if (BG && tid < kNI) {          // this member's record of row t - 1 to every member; the key tables start over
            int it = tid;               // (opaque: addresses made here, not kept in registers across the scans)
            asm volatile("" : "+v"(it));
            const unsigned long long lo = lkey_lo[it], hi = lkey_hi[it];
            const v4u rec = {(unsigned)(lo >> 32), 0xffffffffu - (unsigned)lo, (unsigned)hi, (unsigned)t};
            const int at = ((((t - 1) & 1) * kMaxR + member) * kNI + it) * 16;
            if (local) __builtin_amdgcn_raw_buffer_store_b128(rec, sbuf, at, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(rec, sbuf, at, 0, 16);
            lkey_lo[it] = lkey_hi[it] = gkey_lo[it] = gkey_hi[it] = 0ull;
        }
        if (BAND_ABL & 16) {
            scan(acc, w, tq0, w0, s0, s3, tq_step);
            scan(acc, w, tq0, w0, s4, s5, tq_step);
            scan(acc, w, tq0, w0, s6, s7, tq_step);
            scan(acc, w, tq0, w0, s8, s9, tq_step);
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[e];
            if (sum == 12345.678f) mq[tid] = sum;
            continue;
        }
        // the neighbours' rows t - 1: asked for half-way through the own-row dquads, looked at behind them
        v4u got[kMaxRounds];
        const int par_at = (int)((unsigned)((t - 1) & 1) * xpar);
        scan(acc, w, tq0, w0, s0, s3, tq_step, s1, [&] {
            if (R > 1) {
#pragma unroll
                for (int r = 0; r < kMaxRounds; ++r)
                    if (hx_at[r] >= 0) got[r] = __builtin_amdgcn_raw_buffer_load_b128(xbuf, par_at + hx_at[r], 0, 16);
            }
        });
        BSTAMP(1);
        if (R > 1) {
            const unsigned tag = (unsigned)t;
            unsigned long long since = 0ull;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int r = 0; r < kMaxRounds; ++r)
                    if (hx_at[r] >= 0) ok = ok && (!need0[r] || got[r].y == tag) && (!need1[r] || got[r].w == tag);
                if (__all(ok) || gave_up || (BAND_ABL & 1)) break;
                const unsigned long long now = wall_clock64();
                if (since == 0ull) since = now;
                if (now - since >= ex.wait_ticks) { gave_up = true; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int r = 0; r < kMaxRounds; ++r)
                    if (hx_at[r] >= 0 && !((!need0[r] || got[r].y == tag) && (!need1[r] || got[r].w == tag)))
                        got[r] = __builtin_amdgcn_raw_buffer_load_b128(xbuf, par_at + hx_at[r], 0, 16);
#ifdef BAND_STAMP
                bacc[9] += 1;
#endif
            }
            BSTAMP(2);
#pragma unroll
            for (int r = 0; r < kMaxRounds; ++r) {
                if (need0[r]) *reinterpret_cast<float *>(lds + hw_at[r]) = __uint_as_float(got[r].x);
                if (need1[r]) *reinterpret_cast<float *>(lds + hw_at[r] + 16) = __uint_as_float(got[r].z);
            }
        }
        if (fin) fetch(t);              // the observations of row t + 1 (row t is in `ob` already)
        if (R > 1) {
            if (!(BAND_ABL & 256)) __syncthreads();            // the halo rows t - 1 are in the window
            BSTAMP(3);
            scan(acc, w, tq0, w0, s4, s5, tq_step);
            scan(acc, w, tq0, w0, s6, s7, tq_step);
            scan(acc, w, tq0, w0, s8, s9, tq_step);
            BSTAMP(4);
        }
        if (scans && !(BAND_ABL & 4)) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                __builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float *)(m_at + e * n_own), acc[e], 0, 0, false);
        }
        BSTAMP(5);
        if (BG) {                       // every member's record of row t - 1 (written when this timestep opened there)
            bool lost = false;
            int e0 = tid;
            asm volatile("" : "+v"(e0));
            for (int e = e0; e < R * kNI; e += nthreads) {
                const int m = e / kNI, it = e - m * kNI;
                const int at = ((((t - 1) & 1) * kMaxR + m) * kNI + it) * 16;
                v4u rec;
                unsigned long long since = 0ull;
                for (;;) {
                    rec = __builtin_amdgcn_raw_buffer_load_b128(sbuf, at, 0, 16);
                    if (rec.w == (unsigned)t || gave_up) break;
                    const unsigned long long now = wall_clock64();
                    if (since == 0ull) since = now;
                    if (now - since >= ex.wait_ticks) { lost = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (rec.w == (unsigned)t) {
                    atomicMax(gkey_lo + it, ((unsigned long long)rec.x << 32) | (unsigned long long)(0xffffffffu - rec.y));
                    atomicMax(gkey_hi + it, ((unsigned long long)rec.x << 32) | (unsigned long long)rec.z);
                }
            }
            if (__any(lost)) gave_up = true;
        }
        if (!(BAND_ABL & 512)) __syncthreads();                // every wave is done with the window; M holds the maxima
        BSTAMP(6);
        if (fin) {
            float4 best;
            best.x = fm_at[0];
            best.y = fm_at[4 * n_own];
            best.z = fm_at[8 * n_own];
            best.w = fm_at[12 * n_own];
            fm_at[0] = -INFINITY;
            fm_at[4 * n_own] = -INFINITY;
            fm_at[8 * n_own] = -INFINITY;
            fm_at[12 * n_own] = -INFINITY;
            if (BG) {                   // the candidates from outside the band: fl(largest posterior out there + c)
                int fit_ = fit, fj_ = fj;
                asm volatile("" : "+v"(fit_), "+v"(fj_));
                const unsigned long long glo = gkey_lo[fit_], ghi = gkey_hi[fit_];
                const float rm = unordered((unsigned)(glo >> 32));
                const int pmin = (int)(0xffffffffu - (unsigned)glo), pmax = (int)(unsigned)ghi;
                const float bound = rm + cbg;
                float bs[4] = {best.x, best.y, best.z, best.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = fj_ + k;
                    const bool open = bound >= bs[k];                       // else: nothing outside wins or ties
                    const bool outside = pmin < j - hl || pmax > j + hr;    // a state attaining the row's maximum outside j's band
                    bs[k] = (open && outside) ? fmaxf(bs[k], bound) : bs[k];
                    undecided = undecided || (open && !outside && fin_row && t < flen);
                }
                best = make_float4(bs[0], bs[1], bs[2], bs[3]);
                if (member == 0 && fjg == 0 && t - 1 < flen) bat.rowmax[(size_t)sitem[fit_] * T + t - 1] = rm;      // (the backtrace's bound)
            }
            send(t, settle(best), t + 1 < fmax);
        }
        BSTAMP(7);
        __syncthreads();                // the window holds the own rows t
        BSTAMP(8);
    }
#ifdef BAND_STAMP
    if (lane == 0 && blockIdx.x < 1024)
        for (int i = 0; i < kPhases; ++i) g_phase[((size_t)blockIdx.x * kMaxWaves + wave) * kPhases + i] = bacc[i];
#endif
    if (gave_up && lane == 0) {
        ex.failed[cid] = 1u;
        atomicAdd(&grp.stats[127], 1u);
    }
    nonfinite::raise(odd && fin_row, bat.alarm, grp.serial);
    if (BG) nonfinite::raise(undecided, bat.alarm + (nonfinite::kBandWord - nonfinite::kAlarmWord), grp.serial);
    if (blockIdx.x == 0 && tid == 0) {
        grp.stats[120] = (unsigned)((clock64() - clock_0) >> 4);
        grp.stats[121] = (unsigned)((wall_clock64() - wall_0) >> 4);
    }
}

// The safety net behind a band launch: a tile whose members gave up waiting is decoded again by ONE workgroup, four
// items at a time, posterior rows ping-pong in the LDS, no hand-offs.  grid = tiles, block = 1024, LDS = 32 S bytes.
// (`bg`: the value of every entry outside the band; finite: those candidates are evaluated too, and the rows' maxima go to
// Batch::rowmax for the backtrace)
__global__ __launch_bounds__(1024) void band_repair_kernel(Group grp, const unsigned *__restrict__ failed,
                                                           const float *__restrict__ trans, const float *__restrict__ initial,
                                                           int S, int hl, int hr, float bg) {
    if (failed[blockIdx.x] == 0u) return;
    __shared__ unsigned rowkey[2][4];        // largest entry of the row being made, per item of the group, as an ordered integer
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float4 *rows = reinterpret_cast<float4 *>(lds);
    __shared__ int sframes[kNI], sitem[kNI];
    const int tid = threadIdx.x;
    const int code = grp.tile_map[blockIdx.x];
    const Batch &bat = grp.batch[code >> 20];
    const int B = bat.B, T = bat.T, b0 = (code & 0xfffff) * kNI;
    bool odd = false;
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
    if (tid < 8) rowkey[tid >> 2][tid & 3] = 0u;
    const bool finite_bg = bg != -INFINITY;
    // the row's maxima, for the backtrace's bound (finite background): every thread offers its own, thread bb keeps item bb's
    auto offer = [&](int t, const float (&top)[4]) {
        if (finite_bg)
            for (int bb = 0; bb < 4; ++bb) atomicMax(&rowkey[t & 1][bb], ordered(top[bb]));
    };
    auto keep = [&](int t, int g, const int (&len)[4]) {        // (behind the barrier that closes row t)
        if (finite_bg && tid < 4) {
            if (t < len[tid]) bat.rowmax[(size_t)sitem[4 * g + tid] * T + t] = unordered(rowkey[t & 1][tid]);
            rowkey[t & 1][tid] = 0u;
        }
    };
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
        float top[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int j = tid; j < S; j += 1024) {
            float v[4];
            for (int bb = 0; bb < 4; ++bb) {
                v[bb] = bat.obs[at[bb] + j] + initial[j];
                odd = odd || nonfinite::odd(v[bb]);
                top[bb] = fmaxf(top[bb], v[bb]);
                if (len[bb] > 0) bat.hist[at[bb] + j] = v[bb];
            }
            rows[j] = make_float4(v[0], v[1], v[2], v[3]);
        }
        offer(0, top);
        __syncthreads();
        keep(0, g, len);
        for (int t = 1; t < longest; ++t) {
            const float4 *cur = rows + (size_t)((t - 1) & 1) * S;
            float4 *nxt = rows + (size_t)(t & 1) * S;
            for (int bb = 0; bb < 4; ++bb) top[bb] = -INFINITY;
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
                if (bg != -INFINITY)            // a constant outside the band: every candidate out there, one by one
                    for (int i = 0; i < S; ++i) {
                        if (i >= lo && i <= hi) { i = hi; continue; }
                        const float4 p = cur[i];
                        best[0] = fmaxf(best[0], p.x + bg);
                        best[1] = fmaxf(best[1], p.y + bg);
                        best[2] = fmaxf(best[2], p.z + bg);
                        best[3] = fmaxf(best[3], p.w + bg);
                    }
                float v[4];
                for (int bb = 0; bb < 4; ++bb) {
                    v[bb] = bat.obs[at[bb] + (size_t)t * S + j] + best[bb];
                    odd = odd || nonfinite::odd(v[bb]);
                    top[bb] = fmaxf(top[bb], v[bb]);
                    if (t < len[bb]) bat.hist[at[bb] + (size_t)t * S + j] = v[bb];
                }
                nxt[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
            offer(t, top);
            __syncthreads();
            keep(t, g, len);
        }
    }
    nonfinite::raise(odd, bat.alarm, grp.serial);
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
// (Measured and dropped: asking for the next step's posteriors over [j - 2 hl, j + 2 hr] a step ahead and handing them over
// through the LDS -- 0.64 against 0.36 ms for 512 x 500 x 1440, 1.6 against 0.5 ms for eight batches: the step is not bound
// by the history read, and the wider window is three times the bytes.)
// ---------------------------------------------------------------------------------------
template <int NQ>
struct BandWalker {
    const float *__restrict__ h;          // [T][S] posterior rows of the item
    const float *__restrict__ trans;
    int S, hl, hr, lane;
    float bg;                             // every entry outside the band (-inf, or the constant of band_tile_forward.hpp)
    const float *__restrict__ rowmax;     // [T] largest entry of every posterior row of the item (read when bg is finite)
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
        return lazy::wave_first_argmax4<NQ>(last, lane, S);
    }
    // the state at timestep tt - 1 of the path that is in state j at timestep tt: first argmax inside the band of row j
    __device__ __forceinline__ int step(int j, int tt) const {
        const float4 none = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
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
        // a constant outside the band: no candidate out there exceeds fl(row maximum + constant) (rounding is monotone); where
        // that reaches the band's best, the first argmax is taken over the whole matrix row (the constant is IN the matrix)
        if (bg != -INFINITY && rowmax[tt - 1] + bg >= m) {
            const lazy::RowWalker<NQ> whole{h, trans, S, lane};
            return whole.step(j, tt);
        }
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
        return m == -INFINITY ? 0 : k;          // (every candidate -inf: the reference's scan keeps prev-state 0)
    }
};

template <int NQ>
__global__ __launch_bounds__(64) void group_backtrace_band_kernel(Group grp, const float *__restrict__ trans, int S, int hl,
                                                                  int hr, float bg) {
    const Batch &bat = grp.batch[resident::batch_of_item(grp, blockIdx.x)];
    const int b = (int)blockIdx.x - bat.item0;
    const BandWalker<NQ> w{bat.hist + (size_t)b * bat.T * S, trans, S, hl, hr, (int)threadIdx.x, bg, bat.rowmax + (size_t)b * bat.T};
    lazy::walk_item(w, bat.frames[b], bat.out + (size_t)b * bat.T, bat.T, threadIdx.x);
}

// ... in K speculative segments per item (lazy_backtrace.hpp, chase_segment / stitch_segments): grid = items x K, then items
template <int NQ>
__global__ __launch_bounds__(64) void group_segment_band_kernel(Group grp, const float *__restrict__ trans, int S, int hl, int hr,
                                                                int K, int32_t *__restrict__ arrive, float bg) {
    const int item = (int)blockIdx.x / K, seg = (int)blockIdx.x - item * K;
    const Batch &bat = grp.batch[resident::batch_of_item(grp, item)];
    const int b = item - bat.item0;
    const BandWalker<NQ> w{bat.hist + (size_t)b * bat.T * S, trans, S, hl, hr, (int)threadIdx.x, bg, bat.rowmax + (size_t)b * bat.T};
    lazy::chase_segment(w, bat.frames[b], bat.T, K, seg, bat.out + (size_t)b * bat.T, arrive + (size_t)item * K, threadIdx.x);
}
template <int NQ>
__global__ __launch_bounds__(64) void group_stitch_band_kernel(Group grp, const float *__restrict__ trans, int S, int hl, int hr,
                                                               int K, const int32_t *__restrict__ arrive, float bg) {
    const int item = blockIdx.x;
    const Batch &bat = grp.batch[resident::batch_of_item(grp, item)];
    const int b = item - bat.item0;
    const BandWalker<NQ> w{bat.hist + (size_t)b * bat.T * S, trans, S, hl, hr, (int)threadIdx.x, bg, bat.rowmax + (size_t)b * bat.T};
    lazy::stitch_segments(w, bat.frames[b], bat.T, K, bat.out + (size_t)b * bat.T, arrive + (size_t)item * K, threadIdx.x,
                          grp.stats + 122);
}

// reach of a matrix: *left = max over finite entries of max(j - i, 0), *right = of max(i - j, 0); grid = S, block = 64;
// `reach` set to -1, -1 by the caller -- what a matrix without a finite entry leaves
// (`corner`: what counts as "outside" is the value of trans[0][S - 1], bit for bit -- -inf for log(p), log(tiny) for the
// log(p + tiny) the reference's evaluation decodes with -- instead of -inf; reach[2] <- its bits)
__global__ __launch_bounds__(64) void band_reach_kernel(const float *__restrict__ trans, int32_t *__restrict__ reach, int S,
                                                        int corner) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const float *row = trans + (size_t)j * S;
    const unsigned outside = corner ? __float_as_uint(trans[S - 1]) : __float_as_uint(-INFINITY);
    if (corner && j == 0 && lane == 0) reach[2] = (int32_t)outside;
    const float cv = __uint_as_float(outside);
    int left = -1, right = -1;
    bool below = false;                // an entry BELOW the corner value (the constant's rule wants the band above it)
    for (int i = lane; i < S; i += 64) {
        below = below || row[i] < cv;
        if (__float_as_uint(row[i]) != outside) {
            left = max(left, max(j - i, 0));
            right = max(right, max(i - j, 0));
        }
    }
    if (corner && __any(below) && lane == 0) reach[3] = 1;
    left = -wavered::wave_min_i32(-left);
    right = -wavered::wave_min_i32(-right);
    if (lane == 0 && left >= 0) {
        atomicMax(reach, left);
        atomicMax(reach + 1, right);
    }
}

}  // namespace band
