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

namespace dense {

constexpr int kBT = 64;   // batch items per tile (= posterior panel row width)
constexpr int kNW = 8;    // waves per workgroup = contraction slices
constexpr int kKC = 12;   // prev-state rows staged per chunk

struct Plan {
    int JL;     // next-states per lane (2, 4 or 6); panel row width W = 8*JL
    int W;
    int n_bt;   // batch tiles
    int n_jt;   // state tiles
    int JT;     // next-states per tile (<= W)
    int KS;     // prev-states per wave slice (multiple of kKC)
    int Kp;     // padded contraction length = kNW*KS >= S
};

inline Plan make_plan(int B, int S, int num_cus) {
    Plan best{};
    long best_cost = -1;
    const int n_bt = (B + kBT - 1) / kBT;
    const int cus = num_cus > 0 ? num_cus : 256;
    for (int JL = 6; JL >= 2; JL -= 2) {
        const int W = 8 * JL;
        const int min_jt = (S + W - 1) / W;
        const long tiles = (long)n_bt * min_jt;
        const long rounds = (tiles + cus - 1) / cus;
        const long cost = rounds * W;           // every workgroup computes W slots per round
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            int n_jt = (int)(rounds * cus / n_bt);      // spread the states over the whole round
            if (n_jt < min_jt) n_jt = min_jt;
            if (n_jt > S) n_jt = S;
            best.JL = JL;
            best.W = W;
            best.n_bt = n_bt;
            best.n_jt = n_jt;
            best.JT = (S + n_jt - 1) / n_jt;
            best.n_jt = (S + best.JT - 1) / best.JT;
        }
    }
    const int per_wave = (S + kNW - 1) / kNW;
    best.KS = (per_wave + kKC - 1) / kKC * kKC;
    best.Kp = best.KS * kNW;
    return best;
}

// panel addressing
//   posterior panel  pt[bt][i][64]   : value of batch item 64*bt + s at prev-state i; rows i >= S
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
// t = 0: posterior = obs[b,0,:] + initial  (viterbi.cpp:72-76) into panel 0 and history row 0;
// pad rows of BOTH panels are set to -inf.   grid-stride over n_bt*Kp*64 panel elements.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_panels_kernel(const float *__restrict__ obs,
                                                          const float *__restrict__ initial,
                                                          float *__restrict__ p0, float *__restrict__ p1,
                                                          float *__restrict__ hist, int B, int T, int S,
                                                          int n_bt, int Kp) {
    // one thread per (b, i): reads coalesced along i
    const size_t n = (size_t)n_bt * kBT * Kp;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / Kp);
        const int i = (int)(e - (size_t)b * Kp);
        const int bt = b >> 6, s = b & 63;
        const size_t pe = ((size_t)bt * Kp + i) * kBT + s;
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

// ---------------------------------------------------------------------------------------
// one timestep:  post'[b,j] = obs[b,t,j] + max_i ( post[b,i] + trans[j,i] )     (viterbi.cpp:78-108)
// grid = 8 * ceil(n_bt*n_jt / 8) workgroups of 512 threads; dynamic LDS = lds_bytes(JL)
// ---------------------------------------------------------------------------------------
template <int JL>
struct StepShape {
    static constexpr int W = 8 * JL;
    static constexpr int CHP = kKC * kBT;                 // floats per posterior chunk
    static constexpr int CHT = kKC * W;                   // floats per transition chunk
    static constexpr int NP4 = CHP / 4 / 64;              // float4 per lane per posterior chunk (3)
    static constexpr int NT4 = (CHT / 4 + 63) / 64;       // float4 per lane per transition chunk
    static constexpr int MS = 8 * JL + 4;                 // merge row stride per lane (bank-spread)
    static constexpr int STAGE_FLOATS = kNW * (CHP + CHT);
    static constexpr int MERGE_FLOATS = kNW * 64 * MS;
    static constexpr int LDS_FLOATS = STAGE_FLOATS > MERGE_FLOATS ? STAGE_FLOATS : MERGE_FLOATS;
};

template <int JL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void step_dense_kernel(
    const float *__restrict__ obs, const int32_t *__restrict__ frames,
    const float *__restrict__ trp, const float *__restrict__ pcur, float *__restrict__ pnext,
    float *__restrict__ hist, int B, int T, int S, int t, int n_bt, int n_jt, int JT, int KS, int Kp) {
    using Sh = StepShape<JL>;
    constexpr int W = Sh::W;
    // all LDS is dynamic: a static __shared__ in front would shift the 16-byte alignment the
    // ds_read_b128 fragments rely on
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // XCD-aware tile order: consecutive blocks land on different XCDs (block b -> XCD b % 8), so
    // give each XCD a contiguous run of tiles (state-tile major: its transition panels stay in
    // that XCD's L2 across timesteps)
    const int ntiles = n_bt * n_jt;
    const int per_xcd = (ntiles + 7) >> 3;
    const int L = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (L >= ntiles) return;
    const int jt = L / n_bt, bt = L - jt * n_bt;
    const int b0 = bt * kBT, j0 = jt * JT;

    const int tid = threadIdx.x;
    // skip tiles whose batch items have all ended (t >= batch_frames[b])
    if (!__syncthreads_or(tid < kBT && b0 + tid < B && t < frames[b0 + tid])) return;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bg = lane & 7, jg = lane >> 3;

    float acc[8][JL];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb)
#pragma unroll
        for (int jj = 0; jj < JL; ++jj) acc[bb][jj] = -INFINITY;

    const float4 *gp = reinterpret_cast<const float4 *>(pcur + ((size_t)bt * Kp + (size_t)wave * KS) * kBT);
    const float4 *gt = reinterpret_cast<const float4 *>(trp + ((size_t)jt * Kp + (size_t)wave * KS) * W);
    float *lp = smem + wave * (Sh::CHP + Sh::CHT);
    float *lt = lp + Sh::CHP;
    float4 *lp4 = reinterpret_cast<float4 *>(lp);
    float4 *lt4 = reinterpret_cast<float4 *>(lt);

    const int nch = KS / kKC;
    // staging registers: NP4 + NT4 float4 per lane, the last transition one only on the lanes
    // that have an element (CHT/4 is not a multiple of 64 for W = 48).  Named scalars, no
    // conditional array writes: those would be demoted to scratch memory.
    static_assert(Sh::NP4 == 3 && Sh::NT4 >= 1 && Sh::NT4 <= 3, "staging register layout");
    constexpr bool kTail = (Sh::CHT / 4) % 64 != 0;           // last transition float4 is partial
    const bool tail_lane = lane + 64 * (Sh::NT4 - 1) < Sh::CHT / 4;
    float4 rp0 = gp[lane], rp1 = gp[lane + 64], rp2 = gp[lane + 128];
    float4 rt0 = make_float4(0, 0, 0, 0), rt1 = rt0, rt2 = rt0;
    if (Sh::NT4 > 1 || !kTail || tail_lane) rt0 = gt[lane];
    if (Sh::NT4 > 1 && (Sh::NT4 > 2 || !kTail || tail_lane)) rt1 = gt[lane + 64];
    if (Sh::NT4 > 2 && (!kTail || tail_lane)) rt2 = gt[lane + 128];

    for (int c = 0; c < nch; ++c) {
        // registers -> this wave's private LDS stage (same wave wrote/reads it: program order)
        lp4[lane] = rp0; lp4[lane + 64] = rp1; lp4[lane + 128] = rp2;
        if (Sh::NT4 > 1 || !kTail || tail_lane) lt4[lane] = rt0;
        if (Sh::NT4 > 1 && (Sh::NT4 > 2 || !kTail || tail_lane)) lt4[lane + 64] = rt1;
        if (Sh::NT4 > 2 && (!kTail || tail_lane)) lt4[lane + 128] = rt2;
        if (c + 1 < nch) {
            gp += Sh::CHP / 4;
            gt += Sh::CHT / 4;
            rp0 = gp[lane]; rp1 = gp[lane + 64]; rp2 = gp[lane + 128];
            if (Sh::NT4 > 1 || !kTail || tail_lane) rt0 = gt[lane];
            if (Sh::NT4 > 1 && (Sh::NT4 > 2 || !kTail || tail_lane)) rt1 = gt[lane + 64];
            if (Sh::NT4 > 2 && (!kTail || tail_lane)) rt2 = gt[lane + 128];
        }
#pragma unroll
        for (int ip = 0; ip < kKC; ip += 2) {
            float p0[8], p1[8], t0[JL], t1[JL];
            {
                const float4 a = *reinterpret_cast<const float4 *>(&lp[ip * kBT + 4 * bg]);
                const float4 b = *reinterpret_cast<const float4 *>(&lp[ip * kBT + 32 + 4 * bg]);
                const float4 c4 = *reinterpret_cast<const float4 *>(&lp[(ip + 1) * kBT + 4 * bg]);
                const float4 d = *reinterpret_cast<const float4 *>(&lp[(ip + 1) * kBT + 32 + 4 * bg]);
                p0[0] = a.x; p0[1] = a.y; p0[2] = a.z; p0[3] = a.w;
                p0[4] = b.x; p0[5] = b.y; p0[6] = b.z; p0[7] = b.w;
                p1[0] = c4.x; p1[1] = c4.y; p1[2] = c4.z; p1[3] = c4.w;
                p1[4] = d.x; p1[5] = d.y; p1[6] = d.z; p1[7] = d.w;
            }
#pragma unroll
            for (int q = 0; q < JL / 2; ++q) {
                const float2 u = *reinterpret_cast<const float2 *>(&lt[ip * W + 16 * q + 2 * jg]);
                const float2 v = *reinterpret_cast<const float2 *>(&lt[(ip + 1) * W + 16 * q + 2 * jg]);
                t0[2 * q] = u.x; t0[2 * q + 1] = u.y;
                t1[2 * q] = v.x; t1[2 * q + 1] = v.y;
            }
#pragma unroll
            for (int bb = 0; bb < 8; ++bb)
#pragma unroll
                for (int jj = 0; jj < JL; ++jj)
                    acc[bb][jj] = max3(acc[bb][jj], p0[bb] + t0[jj], p1[bb] + t1[jj]);
        }
    }

    // merge the 8 contraction slices through LDS (aliases the staging area)
    __syncthreads();
    {
        float *m = smem + ((size_t)wave * 64 + lane) * Sh::MS;
#pragma unroll
        for (int bb = 0; bb < 8; ++bb)
#pragma unroll
            for (int jj = 0; jj < JL; ++jj) m[bb * JL + jj] = acc[bb][jj];
    }
    __syncthreads();

    // finalize: wave w owns tile rows (batch positions) 8w .. 8w+7, lane l owns state position l
    if (lane < JT && j0 + lane < S) {
        const int p = lane;
        const int src_jg = (p & 15) >> 1;
        const int jj = 2 * (p >> 4) + (p & 1);
        const int j = j0 + p;
        float out[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = 8 * wave + u;
            const int src_bg = (r & 31) >> 2;
            const int bb = (r & 3) + 4 * (r >> 5);
            const float *m = smem + (size_t)(src_jg * 8 + src_bg) * Sh::MS + bb * JL + jj;
            float v = m[0];
#pragma unroll
            for (int w = 1; w < kNW; ++w) v = fmaxf(v, m[(size_t)w * 64 * Sh::MS]);
            const int b = b0 + r;
            float o = 0.0f;
            if (b < B) {
                const size_t e = ((size_t)b * T + t) * S + j;
                o = obs[e] + v;
                if (t < frames[b]) hist[e] = o;
            }
            out[u] = o;
        }
        float4 *dst = reinterpret_cast<float4 *>(pnext + ((size_t)bt * Kp + j) * kBT + 8 * wave);
        dst[0] = make_float4(out[0], out[1], out[2], out[3]);
        dst[1] = make_float4(out[4], out[5], out[6], out[7]);
    }
}

template <int JL>
constexpr size_t lds_bytes() { return sizeof(float) * (size_t)StepShape<JL>::LDS_FLOATS; }

}  // namespace dense
