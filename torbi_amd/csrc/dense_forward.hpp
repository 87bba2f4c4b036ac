// dense_forward.hpp -- the large-batch forward recurrence for gfx950: one timestep of the whole
// batch as a VALUE-ONLY (max,+) matrix product, plus the posterior history it leaves for the
// lazy backtrace (lazy_backtrace.hpp).
//
// Replaces, for B >= 32, the reference's forward pass (torbi/csrc/viterbi.cpp:65-108, CUDA:
// torbi/csrc/cuda/viterbi.cu:48-130).  Design notes (measurements in DESIGN.md):
//
//  * No backpointers are computed here.  The reference materialises argmax_i for every (b,t,j)
//    (viterbi.cpp:94-100) but its backtrace (viterbi.cpp:153-157) reads exactly one of them per
//    (b,t).  We store the posterior rows instead (same 4*S bytes per timestep as the int32
//    trellis) and recompute the first-argmax only along the decoded path, with the reference's
//    own arithmetic (lazy_backtrace.hpp).  The forward cell is then add + max only.
//  * gfx950 VALU: v_add_f32 issues on either of a SIMD's two 16-lane pipes (2 cycles per wave
//    instruction with >= 2 waves/SIMD); v_max_f32 / v_max3_f32 / v_pk_* occupy the main pipe
//    for 4.  The cheapest exact cell is therefore  add, add, max3  per two prev-states
//    (1.5 instr / cell, measured 39-41 Tcell/s chip-wide); v_pk_add_f32 must be avoided.
//  * Tiling: one workgroup per CU computes a 64-item x JT-state output tile (JT <= 8*JL),
//    8 waves split the contraction (prev-state) axis 8 ways; each lane owns an 8 x JL register
//    tile, operands come from LDS as ds_read_b128/b64 (0.29 dwords per cell at JL = 6).
//    Operand panels are pre-packed so that the LDS image equals the global image (linear copy).
//  * The 8 partial maxima per output are merged through LDS once per timestep.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#ifndef DENSE_ABLATE
#define DENSE_ABLATE 0   // tools/step_bench.hip only: 1 = no global loads after chunk 1, 3 = no merge/finalize
#endif

#ifndef DENSE_TIMING
#define DENSE_TIMING 0    // tools/step_bench.hip only: per-wave s_memtime stamps into dense::timing_buf
#endif

namespace dense {

#if DENSE_TIMING
__device__ unsigned long long timing_buf[4096 * 8];
#define DENSE_STAMP(k) do { if (lane == 0) timing_buf[(blockIdx.x * NW + wave) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define DENSE_STAMP(k) do { } while (0)
#endif


struct Plan {
    int BL;     // batch items per lane (8 or 4); batch tile width BT = 8*BL
    int BT;
    int JL;     // next-states per lane (2, 4 or 6); state panel row width W = 8*JL
    int W;
    int n_bt;   // batch tiles
    int n_jt;   // state tiles
    int JT;     // next-states per tile (<= W)
    int NW;     // waves per workgroup = contraction slices (8 or 16)
    int MSL;    // partial tiles merged through LDS at once (8; 4 = small-footprint variant)
    int KC;     // prev-state rows staged per chunk (12 with 8 waves, 6 with 16)
    int Kp;     // padded contraction length (multiple of KC) >= S
    int NCH;    // chunks of KC prev-state rows per panel = Kp / KC
    int RB;     // XCD region: RB batch tiles x RJ state tiles per XCD (L2 locality only)
};

// Tiling plan: a pure function of (B, S).  One workgroup computes a BT x JT output tile.
//  * BT = 32 doubles the number of workgroups (two co-resident per CU, 4 waves per SIMD): a
//    wave can issue one VALU instruction per ~4 cycles but a SIMD retires two, so stalls of
//    one wave are only hidden when >= 3-4 waves share the SIMD.  BT = 64 halves the L2->LDS
//    operand traffic instead.  `bl_override` (0 = heuristic) exists for experiments.
//  * JL is chosen to minimise rounds * W (every workgroup computes all W state slots).
inline Plan make_plan(int B, int S, int num_cus, int bl_override = 0, int nw_override = 0) {
    Plan best{};
    const int cus = num_cus > 0 ? num_cus : 256;
    const int BL = bl_override ? bl_override : 8;   // measured: 40.6 us/step (BL=8) vs 43.2 (BL=4) at B=512,S=1440
    const int BT = 8 * BL;
    const int per_cu = BL == 4 ? 2 : 1;         // co-resident workgroups per CU
    const int n_bt = (B + BT - 1) / BT;
    long best_cost = -1;
    for (int JL = 6; JL >= 2; JL -= 2) {
        const int W = 8 * JL;
        const int min_jt = (S + W - 1) / W;
        const long tiles = (long)n_bt * min_jt;
        const long slots = (long)cus * per_cu;
        const long rounds = (tiles + slots - 1) / slots;
        const long cost = rounds * W;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            int n_jt = (int)(rounds * slots / n_bt);    // spread the states over the whole round
            if (n_jt < min_jt) n_jt = min_jt;
            if (n_jt > S) n_jt = S;
            best.JL = JL;
            best.W = W;
            best.JT = (S + n_jt - 1) / n_jt;
            best.n_jt = (S + best.JT - 1) / best.JT;
        }
    }
    best.BL = BL;
    best.BT = BT;
    best.n_bt = n_bt;
    // 8 waves x 12-row chunks.  A 16-wave / 6-row variant (4 waves per SIMD, two-round merge) was
    // measured: its waves finish the contraction staggered (mean 48.7K ticks instead of 66.5K) but
    // the workgroup ends no earlier (42.1 vs 40.4 us/step) -- the SIMD's VALU throughput for this
    // instruction mix does not improve with occupancy.  `nw_override` = 16 still selects it.
    best.NW = (nw_override == 16 && BL == 8 && best.JL == 6) ? 16 : 8;
    best.KC = best.NW == 16 ? 6 : 12;
    // (a small-footprint variant -- 8-row chunks, 4 merge slices, 57 KB LDS, 128 VGPRs, two
    // workgroups of two in-flight decodes per CU -- was measured at 44.1 vs 37.3 us per step: the
    // SIMDs are VALU-bound, extra resident waves only add staging overhead)
    best.MSL = 8;
    best.NCH = (S + best.KC - 1) / best.KC;
    best.Kp = best.NCH * best.KC;
    best.RB = n_bt >= 2 ? (n_bt + 1) / 2 : 1;
    return best;
}

// panel addressing
//   posterior panel  pt[bt][i][BT]   : value of batch item BT*bt + s at prev-state i; rows i >= S
//                                      hold -inf (a -inf candidate never raises a maximum)
//   transition panel trp[jt][i][W]   : trans[(jt*JT + s) * S + i]; 0 outside the matrix

// ---------------------------------------------------------------------------------------
// once per decode: pack the transition matrix into per-tile panels (tiled transpose via LDS)
// grid = (ceil(Kp/64), n_jt), block = 256
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_transition_kernel(const float *__restrict__ trans,
                                                              float *__restrict__ trp, int S, int JT,
                                                              int W, int Kp) {
    __shared__ float tile[48][65];
    const int jt = blockIdx.y;
    const int i0 = blockIdx.x * 64;
    const int j0 = jt * JT;
    const int tid = threadIdx.x;
    for (int e = tid; e < W * 64; e += 256) {       // read along i (coalesced)
        const int s = e >> 6, ii = e & 63;
        const int j = j0 + s, i = i0 + ii;
        tile[s][ii] = (s < JT && j < S && i < S) ? trans[(size_t)j * S + i] : 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < W * 64; e += 256) {       // write along s (coalesced)
        const int ii = e / W, s = e - ii * W;
        const int i = i0 + ii;
        if (i < Kp) trp[((size_t)jt * Kp + i) * W + s] = tile[s][ii];
    }
}

// ---------------------------------------------------------------------------------------
// once per decode: for every state tile, the ascending list of chunks (KC prev-state rows) that
// hold at least one transition value other than -inf.  A candidate post[i] + (-inf) = -inf can
// never raise a maximum, so the forward pass skips the other chunks EXACTLY: banded / diagonal /
// sparse transition matrices (e.g. the reference's own pitch transition,
// torbi/evaluate/core.py:24-33) cost only their non-(-inf) blocks; a dense matrix lists every chunk.
// chunks[jt][0] = count, chunks[jt][1..count] = chunk ids.   grid = n_jt, block = 256
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_chunk_lists_kernel(const float *__restrict__ trp,
                                                                int32_t *__restrict__ chunks, int S, int JT,
                                                                int W, int Kp, int NCH, int KC) {
    extern __shared__ int flags[];
    const int jt = blockIdx.x;
    const float *panel = trp + (size_t)jt * Kp * W;
    for (int c = threadIdx.x; c < NCH; c += blockDim.x) {
        int any = 0;
        for (int r = 0; r < KC && !any; ++r) {
            const int i = c * KC + r;
            if (i >= S) break;
            const float *row = panel + (size_t)i * W;
            for (int q = 0; q < JT; ++q)
                if (!(row[q] == -INFINITY)) { any = 1; break; }
        }
        flags[c] = any;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t *list = chunks + (size_t)jt * (NCH + 1);
        int n = 0;
        for (int c = 0; c < NCH; ++c)
            if (flags[c]) list[1 + n++] = c;
        list[0] = n;
    }
}

// ---------------------------------------------------------------------------------------
// t = 0: posterior = obs[b,0,:] + initial  (viterbi.cpp:72-76) into panel 0 and history row 0;
// pad rows of BOTH panels are set to -inf.   grid-stride over n_bt*BT*Kp elements.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_panels_kernel(const float *__restrict__ obs,
                                                          const float *__restrict__ initial,
                                                          float *__restrict__ p0, float *__restrict__ p1,
                                                          float *__restrict__ hist, int B, int T, int S,
                                                          int n_bt, int BT, int Kp) {
    // one thread per (b, i): reads coalesced along i
    const size_t n = (size_t)n_bt * BT * Kp;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / Kp);
        const int i = (int)(e - (size_t)b * Kp);
        const int bt = b / BT, s = b - bt * BT;
        const size_t pe = ((size_t)bt * Kp + i) * BT + s;
        if (i >= S) {
            p0[pe] = -INFINITY;
            p1[pe] = -INFINITY;
        } else {
            float v = 0.0f;
            if (b < B) {
                v = obs[(size_t)b * T * S + i] + initial[i];
                hist[(size_t)b * T * S + i] = v;
            }
            p0[pe] = v;
            p1[pe] = 0.0f;
        }
    }
}

// max(acc, c0, c1) -> one v_max3_f32; the adds stay VOP2 v_add_f32 (the file is compiled with
// -fno-slp-vectorize so that no v_pk_add_f32, half rate on gfx950, is formed)
__device__ __forceinline__ float max3(float a, float b, float c) {
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}

template <int BL, int JL, int NW, int KC, int MSL = 8>
struct StepShape {
    static_assert(KC % 2 == 0 && KC >= 4, "fragments are prev-state pairs, two in flight");
    static constexpr int BT = 8 * BL;
    static constexpr int W = 8 * JL;
    static constexpr int CHP = KC * BT;                   // floats per posterior chunk
    static constexpr int CHT = KC * W;                    // floats per transition chunk
    static constexpr int STAGE = CHP + CHT;               // floats per wave per stage
    static constexpr int MS = BL * JL + 4;                // merge row stride per lane (bank-spread)
    static constexpr int STAGE_FLOATS = NW * 2 * STAGE;   // two stages per wave (ping-pong)
    static constexpr int MERGE_FLOATS = MSL * 64 * MS;    // MSL partial tiles are in LDS at once
    static constexpr int LDS_FLOATS = STAGE_FLOATS > MERGE_FLOATS ? STAGE_FLOATS : MERGE_FLOATS;
};

// operand fragment of one prev-state pair (i, i+1) for this lane's BL x JL register tile
template <int BL, int JL>
struct Frag {
    float p0[BL], p1[BL], t0[JL], t1[JL];
};

template <int BL, int JL>
__device__ __forceinline__ void load_frag(Frag<BL, JL> &f, const float *lp, const float *lt, int ip,
                                          int bg, int jg) {
    constexpr int BT = 8 * BL, W = 8 * JL;
#pragma unroll
    for (int h = 0; h < BL / 4; ++h) {
        const float4 a = *reinterpret_cast<const float4 *>(&lp[ip * BT + 32 * h + 4 * bg]);
        const float4 b = *reinterpret_cast<const float4 *>(&lp[(ip + 1) * BT + 32 * h + 4 * bg]);
        f.p0[4 * h] = a.x; f.p0[4 * h + 1] = a.y; f.p0[4 * h + 2] = a.z; f.p0[4 * h + 3] = a.w;
        f.p1[4 * h] = b.x; f.p1[4 * h + 1] = b.y; f.p1[4 * h + 2] = b.z; f.p1[4 * h + 3] = b.w;
    }
    // state slots of lane jg: JL = 6: {4jg..4jg+3} U {32+2jg, 33+2jg}; JL = 4: {4jg..4jg+3};
    // JL = 2: {2jg, 2jg+1}  -> one ds_read_b128 (+ one ds_read_b64) per row, never ds_read2_b64
    if (JL >= 4) {
        const float4 u = *reinterpret_cast<const float4 *>(&lt[ip * W + 4 * jg]);
        const float4 v = *reinterpret_cast<const float4 *>(&lt[(ip + 1) * W + 4 * jg]);
        f.t0[0] = u.x; f.t0[1] = u.y; f.t0[2] = u.z; f.t0[3] = u.w;
        f.t1[0] = v.x; f.t1[1] = v.y; f.t1[2] = v.z; f.t1[3] = v.w;
    }
    if (JL == 6 || JL == 2) {
        constexpr int off = JL == 6 ? 32 : 0, k = JL == 6 ? 4 : 0;
        const float2 u = *reinterpret_cast<const float2 *>(&lt[ip * W + off + 2 * jg]);
        const float2 v = *reinterpret_cast<const float2 *>(&lt[(ip + 1) * W + off + 2 * jg]);
        f.t0[k] = u.x; f.t0[k + 1] = u.y;
        f.t1[k] = v.x; f.t1[k + 1] = v.y;
    }
}

// 2*BL*JL cells of one fragment: add, add, max3 per prev-state pair.  Orderings that separate a
// v_max3_f32 from the v_add_f32 pair it consumes (skewing by one cell, or issuing the adds of one
// to four batch rows before their max3s with sched_barriers) were measured and are not faster in
// this kernel (contraction loop 66.2K vs 66.5K / 68.2K-70.4K ticks), although a register-only
// probe prefers them (tools/ubench8); the plain order is kept.  Likewise v_min3_u32 on the bit
// patterns (valid when every candidate is <= 0; tools/ubench9 measures add,add,min3_u32 13 % above
// add,add,max3_f32) runs the loop in 68.8K ticks.
template <int BL, int JL>
__device__ __forceinline__ void cells(float (&acc)[BL][JL], const Frag<BL, JL> &f) {
#pragma unroll
    for (int bb = 0; bb < BL; ++bb)
#pragma unroll
        for (int jj = 0; jj < JL; ++jj)
            acc[bb][jj] = max3(acc[bb][jj], f.p0[bb] + f.t0[jj], f.p1[bb] + f.t1[jj]);
}

// One chunk (KC prev-state rows of the posterior panel + of the transition panel) global -> this
// wave's LDS stage by LDS-DMA: the LDS image equals the global image, 1 KiB per wave instruction
// (destination = M0 = wave-uniform LDS byte address, + lane*16 B), no VGPRs, no ds_write.
// Written as inline asm on purpose: hipcc orders every later ds_read behind a pending builtin
// LDS-DMA with s_waitcnt vmcnt(0), which would serialise the copy of chunk c+1 with the cells of
// chunk c.  Here the issuing wave is the only reader of its stage and waits with its own
// s_waitcnt vmcnt(0) one chunk later.  M0 is saved/restored inside the statement.
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}

template <int P4, int T4>
__device__ __forceinline__ void dma_chunk(const float *gp, const float *gt, float *dst, int chp_floats,
                                          int lane) {
    const unsigned base = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)dst);
#pragma unroll
    for (int q = 0; q < (P4 + 63) / 64; ++q) {
        if ((q + 1) * 64 <= P4 || lane + 64 * q < P4) glds16(gp + 4 * (lane + 64 * q), base + 1024 * q);
    }
#pragma unroll
    for (int q = 0; q < (T4 + 63) / 64; ++q) {
        if ((q + 1) * 64 <= T4 || lane + 64 * q < T4)
            glds16(gt + 4 * (lane + 64 * q), base + 4 * chp_floats + 1024 * q);
    }
}

// ---------------------------------------------------------------------------------------
// one timestep:  post'[b,j] = obs[b,t,j] + max_i ( post[b,i] + trans[j,i] )     (viterbi.cpp:78-108)
// grid = 8 * ceil(n_bt*n_jt / 8) workgroups of 64*NW threads; dynamic LDS = lds_bytes<...>()
//
// lane map: bg = lane & 7 (batch group), jg = lane >> 3 (state group); the lane's register tile
// is batch positions {4bg..4bg+3} (+32 for BL = 8) x the state slots listed in load_frag.
// The NW waves split the tile's chunk list NW ways; a chunk (KC prev-state rows) goes global ->
// the wave's private LDS stage by LDS-DMA (two stages, ping-pong) -> ds_read fragments, with two
// fragments (prev-state pairs) in flight while 2*BL*JL cells of a third are computed.
// ---------------------------------------------------------------------------------------
template <int BL, int JL, int NW, int KC, int MSL = 8>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4 * ((BL == 4 || MSL == 4) ? 2 : 1), NW / 4 * ((BL == 4 || MSL == 4) ? 2 : 1))))
void step_dense_kernel(const float *__restrict__ obs, const int32_t *__restrict__ frames,
                       const float *__restrict__ trp, const float *__restrict__ pcur,
                       float *__restrict__ pnext, float *__restrict__ hist,
                       const int32_t *__restrict__ chunks, int B, int T, int S, int t, int n_bt, int n_jt,
                       int JT, int Kp, int NCH, int RB, unsigned *__restrict__ clock_out) {
    using Sh = StepShape<BL, JL, NW, KC, MSL>;
    // (workgroup 0 leaves the shader-clock ticks and the 100 MHz wall-clock ticks of its run behind: the clock the vector
    // ALU's ceiling has to be priced at is the one delivered under this kernel's load, not the 2.4 GHz of the data sheet)
    const unsigned long long clock_0 = clock64(), wall_0 = wall_clock64();
    constexpr int W = Sh::W, BT = Sh::BT;
    constexpr int RW = BT / NW;                     // tile rows (batch positions) finalised per wave
    static_assert(RW % 4 == 0 && RW >= 4, "each wave finalises whole groups of 4 batch rows");
    // all LDS is dynamic: a static __shared__ in front would shift the 16-byte alignment the
    // ds_read_b128 fragments rely on
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // XCD-aware tile order (block b runs on XCD b % 8): XCD x owns a compact RB x RJ rectangle
    // of tiles so that its transition panels and posterior panels stay in that XCD's L2.
    // Pure speed choice; any mapping is correct (inputs come from the previous launch).
    const int ntiles = n_bt * n_jt;
    int bt, jt;
    {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int nrb = (n_bt + RB - 1) / RB;                 // regions along the batch axis
        const int nrj = (8 + nrb - 1) / nrb;                  // regions along the state axis
        const int RJ = (n_jt + nrj - 1) / nrj;
        const bool rect = nrb * nrj == 8 && n_bt % RB == 0 && n_jt % RJ == 0;
        if (rect) {
            const int kb = k % RB, kj = k / RB;
            if (kj >= RJ) return;
            bt = (xcd % nrb) * RB + kb;
            jt = (xcd / nrb) * RJ + kj;
        } else {                                              // uneven grids: linear order
            const int L = blockIdx.x;
            if (L >= ntiles) return;
            jt = L / n_bt;
            bt = L - jt * n_bt;
        }
    }
    const int b0 = bt * BT, j0 = jt * JT;

    const int tid = threadIdx.x;
    // skip tiles whose batch items have all ended (t >= batch_frames[b])
    if (!__syncthreads_or(tid < BT && b0 + tid < B && t < frames[b0 + tid])) return;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bg = lane & 7, jg = lane >> 3;

    // This wave contracts a contiguous 1/NW of the tile's chunk list (build_chunk_lists_kernel):
    // entries [first, first + nch); lane l keeps the id of the wave's l-th chunk.  A dense panel
    // (every chunk listed, or more than 64*NW chunks) is walked arithmetically: the id lookup
    // costs ~1.3 us per launch at 15 chunks per wave.
    const int32_t *list = chunks + (size_t)jt * (NCH + 1);
    const int count = NCH <= 64 * NW ? __builtin_amdgcn_readfirstlane(list[0]) : NCH;
    const bool listed = count != NCH;
    const int per = (count + NW - 1) / NW;
    const int first = wave * per;
    const int nch = count - first < per ? (count - first > 0 ? count - first : 0) : per;
    int ids = 0;
    if (listed && lane < nch) ids = list[1 + first + lane];
#define DENSE_CHUNK_ID(c) (listed ? __builtin_amdgcn_readlane(ids, (c)) : first + (c))

    // observation values of the outputs this lane finalises (wave w: batch rows RW*w .. RW*w+RW-1,
    // lane l: state position l); issued now so that their latency hides under the contraction
    float ob[RW];
    const bool fin_lane = lane < JT && j0 + lane < S;
#pragma unroll
    for (int u = 0; u < RW; ++u) {
        const int b = b0 + RW * wave + u;
        ob[u] = (fin_lane && b < B) ? obs[((size_t)b * T + t) * S + j0 + lane] : 0.0f;
    }

    DENSE_STAMP(0);
    float acc[BL][JL];
#pragma unroll
    for (int bb = 0; bb < BL; ++bb)
#pragma unroll
        for (int jj = 0; jj < JL; ++jj) acc[bb][jj] = -INFINITY;

    const float *gp_base = pcur + (size_t)bt * Kp * BT;
    const float *gt_base = trp + (size_t)jt * Kp * W;
    float *stage = smem + wave * 2 * Sh::STAGE;
    constexpr int P4 = Sh::CHP / 4, T4 = Sh::CHT / 4;         // float4 per chunk
    constexpr int NF = KC / 2;                                // fragments (prev-state pairs) per chunk

    Frag<BL, JL> frag[2];
    DENSE_STAMP(1);
    if (nch > 0) {
        const int ci = DENSE_CHUNK_ID(0);
        dma_chunk<P4, T4>(gp_base + (size_t)ci * Sh::CHP, gt_base + (size_t)ci * Sh::CHT, stage, Sh::CHP, lane);
    }
    for (int c = 0; c < nch; ++c) {
        const float *cur = stage + (c & 1) * Sh::STAGE;
        float *nxt = stage + ((c + 1) & 1) * Sh::STAGE;
        // chunk c has landed (it was issued one whole chunk of cells ago); chunk c+1 -> the other
        // stage, whose previous contents (chunk c-1) were consumed before this point
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c == 0) DENSE_STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        if (DENSE_ABLATE != 1 && c + 1 < nch) {
            const int ci = DENSE_CHUNK_ID(c + 1);
            dma_chunk<P4, T4>(gp_base + (size_t)ci * Sh::CHP, gt_base + (size_t)ci * Sh::CHT, nxt, Sh::CHP, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_frag<BL, JL>(frag[0], cur, cur + Sh::CHP, 0, bg, jg);
        load_frag<BL, JL>(frag[1], cur, cur + Sh::CHP, 2, bg, jg);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            // cells of fragment n while fragment n+1 (and the DMA) is in flight, then refill
            // this register set with fragment n+2; the sched_barriers keep hipcc from sinking
            // the reads next to their first use
            cells<BL, JL>(acc, frag[n & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (n + 2 < NF) load_frag<BL, JL>(frag[n & 1], cur, cur + Sh::CHP, 2 * (n + 2), bg, jg);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef DENSE_CHUNK_ID

    DENSE_STAMP(2);
#if DENSE_ABLATE == 3
    {   // keep the accumulators live, skip merge + finalize
        float sink = 0.f;
#pragma unroll
        for (int bb = 0; bb < BL; ++bb)
#pragma unroll
            for (int jj = 0; jj < JL; ++jj) sink += acc[bb][jj];
        if (sink == 12345.678f) pnext[tid] = sink + ob[0];
        return;
    }
#endif
    // merge the NW contraction slices through LDS (aliases the staging area).  Slot order is
    // state-major (jj*BL + bb): the batch rows one finalising lane needs are then contiguous, so
    // both the writes and the reads are ds_*_b128.  With 16 waves the upper 8 first fold into the
    // lower 8 (at most 8 partial tiles fit in LDS).
    __syncthreads();
    // fold the upper half of the active waves into the lower half until MSL partial tiles remain
#pragma unroll
    for (int active = NW; active > MSL; active /= 2) {
        const int half = active / 2;
        float *m = smem + ((size_t)(wave % half) * 64 + lane) * Sh::MS;
        if (wave >= half && wave < active) {
#pragma unroll
            for (int jj = 0; jj < JL; ++jj)
#pragma unroll
                for (int h = 0; h < BL / 4; ++h)
                    *reinterpret_cast<float4 *>(&m[jj * BL + 4 * h]) =
                        make_float4(acc[4 * h][jj], acc[4 * h + 1][jj], acc[4 * h + 2][jj], acc[4 * h + 3][jj]);
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int jj = 0; jj < JL; ++jj)
#pragma unroll
                for (int h = 0; h < BL / 4; ++h) {
                    const float4 x = *reinterpret_cast<const float4 *>(&m[jj * BL + 4 * h]);
                    acc[4 * h][jj] = fmaxf(acc[4 * h][jj], x.x);
                    acc[4 * h + 1][jj] = fmaxf(acc[4 * h + 1][jj], x.y);
                    acc[4 * h + 2][jj] = fmaxf(acc[4 * h + 2][jj], x.z);
                    acc[4 * h + 3][jj] = fmaxf(acc[4 * h + 3][jj], x.w);
                }
        }
        __syncthreads();
    }
    constexpr int NSL = NW < MSL ? NW : MSL;                  // partial tiles left
    if (wave < NSL) {
        float *m = smem + ((size_t)wave * 64 + lane) * Sh::MS;
#pragma unroll
        for (int jj = 0; jj < JL; ++jj)
#pragma unroll
            for (int h = 0; h < BL / 4; ++h)
                *reinterpret_cast<float4 *>(&m[jj * BL + 4 * h]) =
                    make_float4(acc[4 * h][jj], acc[4 * h + 1][jj], acc[4 * h + 2][jj], acc[4 * h + 3][jj]);
    }
    __syncthreads();
    DENSE_STAMP(3);

    // finalize: wave w owns tile rows (batch positions) RW*w .. RW*w+RW-1, lane l state position l.
    // Row r lives in lane group bg = (r & 31) >> 2 at register row bb = (r & 3) + 4*(r >> 5); for
    // the 4-row group g of this wave (rows RW*w + 4g .. +3) bg and bb & ~3 are constant.
    if (fin_lane) {
        const int p = lane;
        const int src_jg = JL == 2 ? (p >> 1) : (p < 32 ? (p >> 2) : ((p - 32) >> 1));
        const int jj = JL == 2 ? (p & 1) : (p < 32 ? (p & 3) : 4 + (p & 1));
        const int j = j0 + p;
        float out[RW];
#pragma unroll
        for (int g = 0; g < RW / 4; ++g) {
            const int r0 = RW * wave + 4 * g;
            const int src_bg = (r0 & 31) >> 2;
            const int bb0 = 4 * (r0 >> 5);
            const float *m = smem + (size_t)(src_jg * 8 + src_bg) * Sh::MS + jj * BL + bb0;
            float4 v = *reinterpret_cast<const float4 *>(m);
#pragma unroll
            for (int w = 1; w < NSL; ++w) {
                const float4 x = *reinterpret_cast<const float4 *>(m + (size_t)w * 64 * Sh::MS);
                v.x = fmaxf(v.x, x.x); v.y = fmaxf(v.y, x.y); v.z = fmaxf(v.z, x.z); v.w = fmaxf(v.w, x.w);
            }
            out[4 * g] = v.x; out[4 * g + 1] = v.y; out[4 * g + 2] = v.z; out[4 * g + 3] = v.w;
        }
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            const int b = b0 + RW * wave + u;
            const float o = ob[u] + out[u];
            if (b < B && t < frames[b]) hist[((size_t)b * T + t) * S + j] = o;
            out[u] = b < B ? o : 0.0f;
        }
        float4 *dst = reinterpret_cast<float4 *>(pnext + ((size_t)bt * Kp + j) * BT + RW * wave);
#pragma unroll
        for (int h = 0; h < RW / 4; ++h)
            dst[h] = make_float4(out[4 * h], out[4 * h + 1], out[4 * h + 2], out[4 * h + 3]);
    }
    DENSE_STAMP(4);
    if (clock_out && blockIdx.x == 0 && threadIdx.x == 0) {
        clock_out[0] = (unsigned)(clock64() - clock_0);
        clock_out[1] = (unsigned)(wall_clock64() - wall_0);
    }
}

template <int BL, int JL, int NW, int KC, int MSL = 8>
constexpr size_t lds_bytes() { return sizeof(float) * (size_t)StepShape<BL, JL, NW, KC, MSL>::LDS_FLOATS; }

}  // namespace dense
