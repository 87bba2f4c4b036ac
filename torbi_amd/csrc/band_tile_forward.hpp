// band_tile_forward.hpp -- the band recurrence of band_forward.hpp with ONE workgroup per 16-item tile: every next-state
// of the tile in one workgroup, the time loop inside the launch, no hand-off between workgroups at all.
//
// Where band_forward.hpp splits a tile over R workgroups so that each member's slab of the band fits its LDS (and pays for
// it with halo granules, tickets, bounded waits and three barriers per timestep), a launch group that has a tile for
// (most) compute units needs no split: the workgroup keeps the tile's whole window of the previous posterior row in the LDS
//     W   [4 item groups][S + 4 Dq - 1 rows][4 items]                         103.5 KB at 1440 states, reach 87
// and STREAMS the band.  The band is packed once per launch, diagonal-major in the order the lanes read it,
//     tpack [64-next block][dquad q][16 groups of four next-states][4 diagonals][4 next]     1 KiB per (block, dquad)
// (1.0 MB at 1440 states, reach 87: it stays in every XCD's 4 MB L2, where the 32 workgroups of the XCD read the same
// bytes), and a wave copies the KiB of its next dquads by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write)
// into a ring of four 1 KiB stages of its own -- three dquads ahead, so that a stage has ~2 dquads (500 - 1 500 cycles) to
// land -- and reads it back with the same ds_read_b128 broadcasts as the split kernel reads its slab: per dquad and lane
// 4 x 16 B of band + 4 x 16 B of window for 64 cells (64 v_add_f32 + 32 v_max3_f32); 16 B per compute unit and clock
// from the L2 at the vector ALU's rate.  The issuing wave is the only reader of its ring: its own s_waitcnt vmcnt(2)
// (exactly two younger copies are in flight at every dquad) is the whole synchronisation.
//
// A lane owns 4 next-states x 4 items (as in the split kernel) for ALL diagonals of a 64-next block, so the waves share no
// outputs: no merge buffer, no atomics.  A wave scans its blocks one after the other (wave w: blocks w, w + waves), keeps
// the finished values in registers, and the timestep ends with
//     barrier (every wave is done with the window), own rows -> window, barrier
// -- two barriers per timestep of ~36 us, nothing else between workgroups or waves.  The memory side rides inside the scans:
// a block's observations of row t are asked for, and its history row t - 1 is stored FROM THE WINDOW (it is still there),
// in the middle of the block's scan of timestep t, each wave of a SIMD at another dquad.  Arithmetic and results are those
// of band_forward.hpp (viterbi.cpp:81-104 over the band: -inf candidates never win the strict '>').
// What binds it (profiles/r06_band_tile_ablations.txt): vector-ALU issue at the clock the power budget delivers -- the
// dquads alone take 34.7 us per timestep at 2.10 GHz, the whole launch 36.3 at 2.34.
//
// A CONSTANT outside the band.  The reference's own evaluation does not decode with log(p) but with log(p + tiny)
// (torbi/evaluate/core.py:97-103 -> torbi/core.py:341-347): its pitch matrix is log(tiny) = -87.34 outside the band, not
// -inf, and a candidate from outside the band does win when a row's posteriors fall that far (clamped posteriorgram tails).
// With every entry outside the band equal to ONE value c, the candidates from outside are fl(post[i] + c), and rounding is
// monotone: their maximum is fl(M + c) with M the largest posterior outside the band.  The workgroup keeps, per item, the
// previous row's maximum and the lowest / highest state that attains it (LDS atomics in the finish it runs anyway).  For an
// output (j, item):   fl(rowmax + c) < best of the band          -> nothing outside can win or tie: done (real data: always);
//                     else, a state attaining rowmax outside j's band -> M = rowmax: best = max(best, fl(rowmax + c)), exact;
//                     else (the maximum inside the band, yet not enough: in-band entries below c) -> the tile's batch raises
//                     an alarm and nonfinite::repair_kernel decodes it in the reference's order (never on a pitch matrix).
// The backtrace (band_forward.hpp, BandWalker) makes the same test with the row maxima this kernel leaves in Batch::rowmax
// and scans the whole matrix row for the steps where the band alone does not decide.
#pragma once

#include "band_forward.hpp"

namespace band {

constexpr int kTileWaves = 12;           // at most (three per SIMD: 168 registers a lane; eight waves: 256)
constexpr int kTileBlocks = 3;           // 64-next blocks per wave, at most
constexpr int kTileSlots = 24;           // waves x blocks per wave, at most
constexpr int kRing = 4;                 // stages of the band ring per wave, 1 KiB each

struct TilePlan {
    int S, hl, hr;
    int Dq, Dq4;             // dquads of the band; rounded up to whole turns of the ring (the tail is never evaluated)
    int n_jg, nblk;          // groups of four next-states, 64-next blocks
    int waves, bpw;          // waves per workgroup; blocks per wave
    int w_rows, ig_stride;   // rows of the window (row w holds state w - hl); floats between the item groups' windows
    int w_off, ring_off, misc_off, lds_bytes;
    float background;        // every entry outside the band (-inf: the band kernels' original contract)
};

__host__ __device__ inline size_t tile_pack_bytes(int nblk, int Dq4) { return (size_t)nblk * Dq4 * 1024; }
// what a workspace sets aside for the packed band of any reach the plan accepts
inline size_t tile_pack_bytes_max(int S) {
    if (S % 4 != 0 || S > 64 * kTileSlots) return 0;
    return tile_pack_bytes((S / 4 + 15) / 16, kMaxWindow / 4);
}

// `waves_wanted`: 0, or 8 / 12 for shapes of more than 8 blocks (experiments)
inline bool make_tile_plan(int S, int hl, int hr, TilePlan &p, int waves_wanted = 0) {
    if (S < 64 || S % 4 != 0 || hl < 0 || hr < 0 || hl >= S || hr >= S) return false;
    if (hl + hr + 4 > kMaxWindow) return false;
    p.S = S; p.hl = hl; p.hr = hr;
    p.background = -INFINITY;
    p.Dq = (hl + hr + 1 + 3) / 4;
    p.Dq4 = (p.Dq + 3) / 4 * 4;
    p.n_jg = S / 4;
    p.nblk = (p.n_jg + 15) / 16;
    if (p.nblk > kTileSlots) return false;
    p.w_rows = S + 4 * p.Dq - 1;
    p.ig_stride = 4 * p.w_rows;
    while (p.ig_stride % 64 != 4) p.ig_stride += 4;
    p.w_off = 0;
    p.ring_off = p.w_off + 4 * p.ig_stride * 4;
    // twelve waves (three per SIMD) where the rings fit beside the window, else eight (wide bands)
    for (int waves = p.nblk <= 4 ? 4 : p.nblk <= 8 ? 8 : kTileWaves; waves >= 4; waves -= 4) {
        if (p.nblk > 8 && (waves_wanted == 8 || waves_wanted == 12) && waves > waves_wanted) continue;
        p.waves = waves;
        p.bpw = (p.nblk + waves - 1) / waves;
        if (p.bpw > (waves == 12 ? 2 : 3)) return false;
        p.misc_off = p.ring_off + waves * kRing * 1024;
        p.lds_bytes = p.misc_off + 768;          // frames, offsets, items; row maxima and where they are attained (two rows)
        if (p.lds_bytes <= kLdsBytes) return true;
    }
    return false;
}

// the band, packed: grid = nblk * Dq4, block = 64 (lane = group of four next-states x diagonal of the dquad)
__global__ __launch_bounds__(64) void pack_band_kernel(const float *__restrict__ trans, float *__restrict__ tpack, int S, int hl,
                                                       int hr, int Dq, int Dq4) {
    const int blk = (int)blockIdx.x / Dq4, q = (int)blockIdx.x - blk * Dq4, lane = threadIdx.x;
    const int jgl = lane >> 2, dd = 4 * q + (lane & 3);
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = 4 * (16 * blk + jgl) + k, i = j + dd - hl;
        const bool in = q < Dq && j < S && i >= 0 && i < S && dd <= hl + hr;
        v[k] = in ? trans[(size_t)j * S + i] : -INFINITY;
    }
    reinterpret_cast<float4 *>(tpack)[(size_t)blockIdx.x * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
}

#ifndef BAND_TILE_ABL
#define BAND_TILE_ABL 0       // build-time ablations (timing only, results wrong): 1 no band copies (the ring keeps what it has),
                              // 2 no history stores, 4 no observation loads, 8 no barriers, 16 no dquads
#endif

// 1 KiB, global -> LDS: lane l's 16 bytes from `base + voff` to LDS byte address `dst + 16 l` (dst wave-uniform, through M0)
__device__ __forceinline__ void glds16s(const char *base, unsigned voff, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(dst)
                 : "memory");
}

#ifdef BAND_STAMP
#define TSTAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); bacc[i] += now_ - blast; blast = now_; }
#else
#define TSTAMP(i)
#endif

// grid = tiles of the group, block = 64 * pl.waves, dynamic LDS = pl.lds_bytes
// BG: ONE constant outside the band instead of -inf (an instance of its own: the -inf instance carries none of its code)
template <int BPW, int NW, bool BG = false>
__global__ __launch_bounds__(64 * NW) void band_tile_kernel(Group grp, TilePlan pl, const float *__restrict__ tpack,
                                                                    const float *__restrict__ initial) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *const wq = reinterpret_cast<float *>(lds + pl.w_off);
    int *const sframes = reinterpret_cast<int *>(lds + pl.misc_off);                                 // [16] frames per item (0 past the batch)
    unsigned long long *const sbase = reinterpret_cast<unsigned long long *>(lds + pl.misc_off + 64);  // [16] element offset of the item
    int *const sitem = reinterpret_cast<int *>(lds + pl.misc_off + 192);                             // [16] item numbers
    float *const srm = reinterpret_cast<float *>(lds + pl.misc_off + 256);       // [2 rows by parity][16] largest posterior of the row
    int *const spmin = reinterpret_cast<int *>(lds + pl.misc_off + 384);         // [2][16] lowest state that attains it
    int *const spmax = reinterpret_cast<int *>(lds + pl.misc_off + 512);         // [2][16] highest
    const float cbg = pl.background;
    constexpr bool bg = BG;                    // (a constant outside the band: see the head of the file)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = 64 * pl.waves;
    const unsigned long long clock_0 = clock64(), wall_0 = wall_clock64();      // (stats[120], [121]: the clock under this load)
    const int S = pl.S, hl = pl.hl, Dq = pl.Dq, Dq4 = pl.Dq4, n_jg = pl.n_jg;

    const int code = grp.tile_map[blockIdx.x];
    const int bk = code >> 20, tile = code & 0xfffff;
    const Batch &bat = grp.batch[bk];
    const float *__restrict__ obs = bat.obs;
    float *__restrict__ hist = bat.hist;
    const int B = bat.B, T = bat.T;
    const int b0 = tile * kNI;

    if (tid < kNI) {
        int f = 0;
        const int item = bat.order[b0 + tid < B ? b0 + tid : B - 1];
        if (b0 + tid < B) {
            f = bat.frames[item];
            f = f < 1 ? 1 : (f > T ? T : f);
        }
        sframes[tid] = f;
        sbase[tid] = (unsigned long long)item * (unsigned long long)T * (unsigned long long)S;
        sitem[tid] = item;
    }
    if (tid < 2 * kNI) {
        srm[tid] = -INFINITY;
        spmin[tid] = 0x7fffffff;
        spmax[tid] = -1;
    }
    // rows outside the matrix stay 0: their band entries are -inf
    for (int e = tid; e < 4 * pl.ig_stride; e += nthreads) wq[e] = 0.0f;
    __syncthreads();
    int fmax = 0;
#pragma unroll
    for (int it = 0; it < kNI; ++it) fmax = max(fmax, sframes[it]);

    // ---- the lane: 4 next-states (group jg) x 4 items (group ig) of each of the wave's blocks ----------------------------
    const int ig = lane & 3, jgl = lane >> 2;
    int jg[BPW];
    bool rowok[BPW];
    int nb = 0;
#pragma unroll
    for (int u = 0; u < BPW; ++u) {
        const int blk = wave + u * pl.waves;
        const bool has = blk < pl.nblk;
        nb += has ? 1 : 0;
        const int raw = 16 * blk + jgl;
        rowok[u] = has && raw < n_jg;
        jg[u] = raw < n_jg ? raw : n_jg - 1;
    }
    nb = __builtin_amdgcn_readfirstlane(nb);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    const unsigned ring_m0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)pl.ring_off + (unsigned)wave * (kRing * 1024u));
    const char *const ring_lane = lds + pl.ring_off + wave * (kRing * 1024) + jgl * 64;
    const char *const tp_bytes = reinterpret_cast<const char *>(tpack);
    const unsigned voff = (unsigned)lane * 16u;

    // the band ring: copies are issued in the order the dquads are evaluated, three ahead, for ever
    const unsigned blk_bytes = (unsigned)Dq4 * 1024u;            // a block's dquads in the packed band
    unsigned d_off = (unsigned)wave * blk_bytes;
    int d_q = 0, d_u = 0;
    auto copy_next = [&](unsigned stage) {
        if (!(BAND_TILE_ABL & 1)) glds16s(tp_bytes + d_off, voff, ring_m0 + stage * 1024u);
        ++d_q;
        d_off += 1024u;
        if (d_q == Dq4) {
            d_q = 0;
            d_u = d_u + 1 == nb ? 0 : d_u + 1;
            d_off = (unsigned)(wave + d_u * pl.waves) * blk_bytes;
        }
    };
    auto landed = [&]() {          // all but the two youngest copies have landed
        if (!(BAND_TILE_ABL & 1)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    };
    auto barrier = [&]() {         // (not __syncthreads(): its fence would wait for the copies in flight)
        if (!(BAND_TILE_ABL & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    // post'[j] = obs[t][j] + max (viterbi.cpp:102); row 0 = obs[b][0][:] + initial (viterbi.cpp:72-76).
    // A block's observations of row t are asked for, and its history row t - 1 is stored, in the MIDDLE of the block's scan
    // of timestep t, each wave of a SIMD at another dquad (the row is still in the window: nobody writes the window during
    // the scans).  A workgroup moves 2 x 92 KB per timestep, which the compute unit's memory pipe takes 2 - 3 us to issue:
    // at the end of the timestep -- every wave at once, nothing else running -- that was 10 % of the launch.  Here the
    // wave that has just issued them waits at its next `landed()` (the counter is in order) while the SIMD's other two
    // waves keep the vector ALU busy.
    float4 ob[4] = {};              // observations of the block being scanned: [item] x 4 next-states
    float v[BPW][16];               // the finished rows of this timestep: [block][4 next + item]
    // (the items' offsets and lengths are read from the LDS where they are used: twelve registers less across the scans)
    auto ask = [&](int j4, int t) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (!(BAND_TILE_ABL & 4)) ob[c] = *reinterpret_cast<const float4 *>(obs + sbase[4 * ig + c] + (size_t)t * S + j4);
    };
    auto store_row = [&](int j4, bool ok, int t, const float (&x)[16]) {
        if (BAND_TILE_ABL & 2) return;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (ok && t < sframes[4 * ig + c])
                *reinterpret_cast<float4 *>(hist + sbase[4 * ig + c] + (size_t)t * S + j4) = make_float4(x[c], x[4 + c], x[8 + c], x[12 + c]);
    };
    auto store_from_window = [&](int j4, bool ok, int t) {         // row t of the lane's 4 next-states x 4 items, as the window holds it
        float x[16];
        const float *at = wq + ig * pl.ig_stride + 4 * (hl + j4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 r = *reinterpret_cast<const float4 *>(at + 4 * k);
            x[4 * k] = r.x; x[4 * k + 1] = r.y; x[4 * k + 2] = r.z; x[4 * k + 3] = r.w;
        }
        store_row(j4, ok, t, x);
    };
    bool odd = false;               // a NaN / +inf posterior value was produced (nonfinite.hpp)
    bool undecided = false;         // a constant outside the band, and neither test of the head of the file decided an output
    auto finish = [&](int u, float (&acc)[16], int t) {
        if (bg && t > 0) {          // the candidates from outside the band: fl(largest posterior out there + c)
            const int par = (t - 1) & 1;
            int ig_ = ig, j0_ = 4 * jg[u];      // (opaque: the addresses and band edges below are made HERE, not kept in
            asm volatile("" : "+v"(ig_), "+v"(j0_));      //  registers across the scans)
#pragma unroll
            for (int c = 0; c < 4; ++c) {          // (an item at a time: four words live, not sixteen)
                const int it = par * kNI + 4 * ig_ + c;
                const float bound = srm[it] + cbg;
                const int lo = spmin[it], hi = spmax[it];
                const bool live = t < sframes[4 * ig_ + c];             // (rows past an item's length are nobody's)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0_ + k;
                    const bool open = bound >= acc[4 * k + c];                      // else: nothing outside wins or ties
                    const bool outside = lo < j - hl || hi > j + pl.hr;             // the row's maximum stands outside j's band
                    acc[4 * k + c] = (open && outside) ? fmaxf(acc[4 * k + c], bound) : acc[4 * k + c];
                    undecided = undecided || (open && !outside && rowok[u] && live);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            v[u][0 + c] = ob[c].x + acc[0 + c];
            v[u][4 + c] = ob[c].y + acc[4 + c];
            v[u][8 + c] = ob[c].z + acc[8 + c];
            v[u][12 + c] = ob[c].w + acc[12 + c];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) odd = odd || (rowok[u] && nonfinite::odd(v[u][e]));
        if (bg && rowok[u]) {       // the largest posterior of row t per item (where it is attained: close_timestep)
            int ig_ = ig;
            asm volatile("" : "+v"(ig_));
#pragma unroll
            for (int c = 0; c < 4; ++c)
                __builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float *)(srm + (t & 1) * kNI + 4 * ig_ + c),
                                          fmaxf(fmaxf(v[u][c], v[u][4 + c]), fmaxf(v[u][8 + c], v[u][12 + c])), 0, 0, false);
        }
    };
    auto close_timestep = [&](int t) {
        if (t + 1 >= fmax) return;          // (the last row's maxima are nobody's input)
        barrier();                      // every wave is done with the window of row t - 1 (and the row's maxima are complete)
        if (bg) {
            const int par = t & 1;
            int ig_ = ig;
            asm volatile("" : "+v"(ig_));
            const float4 rm = *reinterpret_cast<const float4 *>(srm + par * kNI + 4 * ig_);
            const float top[4] = {rm.x, rm.y, rm.z, rm.w};
#pragma unroll
            for (int u = 0; u < BPW; ++u)
                if (rowok[u]) {
                    int j0_ = 4 * jg[u];
                    asm volatile("" : "+v"(j0_));
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (v[u][4 * k + c] == top[c]) {
                                atomicMin(spmin + par * kNI + 4 * ig_ + c, j0_ + k);
                                atomicMax(spmax + par * kNI + 4 * ig_ + c, j0_ + k);
                            }
                }
            if (tid < kNI) {            // the row maximum for the backtrace; the other parity's words for row t + 1
                if (t < sframes[tid]) bat.rowmax[(size_t)sitem[tid] * T + t] = srm[par * kNI + tid];
                srm[(par ^ 1) * kNI + tid] = -INFINITY;
                spmin[(par ^ 1) * kNI + tid] = 0x7fffffff;
                spmax[(par ^ 1) * kNI + tid] = -1;
            }
        }
#pragma unroll
        for (int u = 0; u < BPW; ++u)
            if (rowok[u]) {
                float *at = wq + ig * pl.ig_stride + 4 * (hl + 4 * jg[u]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<float4 *>(at + 4 * k) = make_float4(v[u][4 * k], v[u][4 * k + 1], v[u][4 * k + 2], v[u][4 * k + 3]);
            }
        barrier();                      // the window holds row t
    };

#pragma unroll
    for (int u = 0; u < BPW; ++u) {
        if (u >= nb) continue;
        ask(4 * jg[u], 0);
        float first[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = initial[4 * jg[u] + k];
#pragma unroll
            for (int c = 0; c < 4; ++c) first[4 * k + c] = x;
        }
        finish(u, first, 0);
    }
    if (nb > 0 && fmax > 1)
        for (int r = 0; r < kRing - 1; ++r) copy_next((unsigned)r);
    close_timestep(0);

    // where in a block's scan this wave talks to the memory: the waves of a SIMD (w, w + 4, w + 8) a third of the scan apart
    const int nq = Dq4 / 4;
    const int q_ask = 4 * (((wave >> 2) % 3) * nq / 3);
    const int q_store = 4 * min(nq - 1, q_ask / 4 + (nq + 5) / 6);
#ifdef BAND_STAMP
    unsigned long long bacc[kPhases] = {};
    unsigned long long blast = __builtin_readcyclecounter();
#endif
    for (int t = 1; t < fmax; ++t) {
#pragma unroll
        for (int u = 0; u < BPW; ++u) {
            if (u >= nb) continue;
            float acc[16];
            float4 w[8], tt[4];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = -INFINITY;
            const char *const w0 = lds + pl.w_off + (ig * pl.ig_stride + 16 * jg[u]) * 4;
            landed();                   // the block's first stage
#pragma unroll
            for (int d = 0; d < 4; ++d) tt[d] = lds_f4(ring_lane + 16 * d);
#pragma unroll
            for (int m = 0; m < 7; ++m) w[m] = lds_f4(w0 + 16 * m);
            TSTAMP(0);
            for (int q = 0; q < Dq4; q += 4) {
                const char *const wp = w0 + (size_t)q * 64;
                if (q == q_ask) ask(4 * jg[u], t);
                if (q == q_store) store_from_window(4 * jg[u], rowok[u], t - 1);
                copy_next(3u);
                landed();
                if (!(BAND_TILE_ABL & 16)) dquad<0>(acc, w, tt, ring_lane + 1024, wp + 64);
                copy_next(0u);
                landed();
                if (q + 1 < Dq && !(BAND_TILE_ABL & 16)) dquad<1>(acc, w, tt, ring_lane + 2048, wp + 128);
                copy_next(1u);
                landed();
                if (q + 2 < Dq && !(BAND_TILE_ABL & 16)) dquad<0>(acc, w, tt, ring_lane + 3072, wp + 192);
                copy_next(2u);
                landed();
                if (q + 3 < Dq && !(BAND_TILE_ABL & 16)) dquad<1>(acc, w, tt, ring_lane, wp + 256);
            }
            TSTAMP(1);
            finish(u, acc, t);
        }
        TSTAMP(2);
        close_timestep(t);
        TSTAMP(3);
    }
    // the last row never reaches the window: from the registers
#pragma unroll
    for (int u = 0; u < BPW; ++u)
        if (u < nb) store_row(4 * jg[u], rowok[u], fmax - 1, v[u]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (no copy may land in an LDS that is no longer this workgroup's)
    nonfinite::raise(odd, bat.alarm, grp.serial);
    nonfinite::raise(undecided, bat.alarm + (nonfinite::kBandWord - nonfinite::kAlarmWord), grp.serial);
#ifdef BAND_STAMP
    if (lane == 0 && blockIdx.x < 1024)
        for (int i = 0; i < kPhases; ++i) g_phase[((size_t)blockIdx.x * kMaxWaves + wave) * kPhases + i] = bacc[i];
#endif
    if (blockIdx.x == 0 && tid == 0) {
        grp.stats[120] = (unsigned)((clock64() - clock_0) >> 4);
        grp.stats[121] = (unsigned)((wall_clock64() - wall_0) >> 4);
    }
}

}  // namespace band
