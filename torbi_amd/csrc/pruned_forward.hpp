// pruned_forward.hpp -- EXACT forward recurrence that does not evaluate every (prev, next) cell.
//
// The transition matrix is the same for every batch item and timestep, so each row is sorted ONCE
// per decode (descending).  For item b and next state j
//     m = max_i fl(post[b,i] + trans[j,i])                                   (viterbi.cpp:81-104)
// is found by (1) seeding `best` with the R largest posteriors of the item (explicit candidates) and
// (2) scanning row j in descending transition order.  Every candidate not yet examined has
// post <= thr (the (R+1)-th largest posterior) and trans <= t_k (the current list entry), hence
// fl(post + trans) <= fl(thr + t_k) by monotonicity of rounding; once that bound is <= best the
// maximum is final.  Only VALUES are needed here (the backpointer is recomputed along the decoded
// path by lazy_backtrace.hpp), so ties need no care and the result is bit-identical to the dense
// scan.  On the uniform-random benchmark ~8 % of the cells are examined; on peaked posteriors or
// banded matrices far fewer.  Worst case (flat rows) every cell is examined at a higher cost per
// cell than the dense kernel -- the host selects the path (torbi_hip.hip).
//
// Lanes: 16 batch items x 4 next-states per wave.  The 16 lanes of a next-state load 16 consecutive
// list entries with one coalesced 8-byte load each and consume them by DPP row rotation (max is
// order independent): per candidate one DPP address add, one LDS gather from the [prev][16 items]
// posterior tile, one DPP add and half a v_max3.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "wave_reduce.hpp"

namespace pruned {

constexpr int kR = 5;        // explicit top candidates per item; thr = (kR+1)-th largest posterior
constexpr int kNB = 16;      // batch items per tile (= lanes per next-state)
constexpr int kLook = 4;     // 16-entry list blocks in flight per next-state
constexpr int kWaves = 16;   // waves per workgroup
constexpr int kTop = kR + 1;
constexpr int kMaxJT = 16;   // state tiles per batch tile (kMaxJT * kTop candidates <= 2 per lane)

struct Plan {
    int n_bt;    // batch tiles of 16 items
    int n_jt;    // next-state tiles
    int JT;      // next-states per tile (multiple of 4)
    int Sp;      // list length rounded up to 16
    int SpP;     // list row stride in entries: Sp + 16*kLook all-(-inf) entries so prefetch never leaves the row
    int NPOW;    // sort width (power of two >= S)
};

inline bool supported(int B, int S) { return B >= 32 && S % 4 == 0 && S >= 64 && S <= 2048; }

// dynamic LDS of step_pruned_kernel: posterior tile [S][16] + this tile's outputs [16][JT] + merged top lists
inline size_t lds_bytes(int S, int JT) { return sizeof(float) * ((size_t)kNB * S + (size_t)kNB * JT + 2 * kNB * kTop); }

inline Plan make_plan(int B, int S, int num_cus) {
    Plan p{};
    p.n_bt = (B + kNB - 1) / kNB;
    int n_jt = num_cus / p.n_bt;
    if (n_jt < 1) n_jt = 1;
    if (n_jt < (S + 255) / 256) n_jt = (S + 255) / 256;   // the per-tile top selection holds 4 outputs per lane
    if (n_jt > kMaxJT) n_jt = kMaxJT;     // the per-item top lists of all state tiles are merged by one wave
    int JT = (S + n_jt - 1) / n_jt;
    JT = (JT + 3) / 4 * 4;                // S <= 2048 and n_jt >= S/256 keep JT <= 256
    p.JT = JT;
    p.n_jt = (S + JT - 1) / JT;
    p.Sp = (S + 15) / 16 * 16;
    p.SpP = p.Sp + 16 * kLook;
    p.NPOW = 64;
    while (p.NPOW < S) p.NPOW *= 2;
    return p;
}

// ---------------------------------------------------------------------------------------
// once per decode: sort every transition row in descending order (bitonic, one workgroup per row).
// Entry = {t, byte offset of prev-state i in the posterior tile = i * 64}.  grid = S, block = 256,
// dynamic LDS = NPOW * 8 bytes.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sort_rows_kernel(const float *__restrict__ trans,
                                                        float2 *__restrict__ sorted, int S, int SpP, int NPOW) {
    extern __shared__ float skey[];
    int *sval = reinterpret_cast<int *>(skey + NPOW);
    const int j = blockIdx.x;
    const float *row = trans + (size_t)j * S;
    for (int k = threadIdx.x; k < NPOW; k += 256) {
        skey[k] = k < S ? row[k] : -INFINITY;
        sval[k] = k < S ? k * (kNB * 4) : 0;
    }
    __syncthreads();
    for (int size = 2; size <= NPOW; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int k = threadIdx.x; k < NPOW; k += 256) {
                const int partner = k ^ stride;
                if (partner > k) {
                    const bool desc = (k & size) == 0;          // descending in even blocks
                    const float a = skey[k], b = skey[partner];
                    if (desc ? (a < b) : (a > b)) {
                        skey[k] = b; skey[partner] = a;
                        const int t = sval[k]; sval[k] = sval[partner]; sval[partner] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
    float2 *out = sorted + (size_t)j * SpP;
    for (int k = threadIdx.x; k < SpP; k += 256) {
        float2 v;
        v.x = k < S ? skey[k] : -INFINITY;
        v.y = __builtin_bit_cast(float, k < S ? sval[k] : 0);
        out[k] = v;
    }
}

// once per decode: tt[i][j] = trans[j][i] (seed candidates are read along next-states)
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ trans, float *__restrict__ tt,
                                                        int S) {
    __shared__ float tile[32][33];
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (y0 + r < S && x0 + tx < S) ? trans[(size_t)(y0 + r) * S + x0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (x0 + r < S && y0 + tx < S) tt[(size_t)(x0 + r) * S + y0 + tx] = tile[tx][r];
}

// t = 0: history row 0 = obs[b,0,:] + initial                                 (viterbi.cpp:72-76)
__global__ __launch_bounds__(256) void init_history_kernel(const float *__restrict__ obs,
                                                           const float *__restrict__ initial,
                                                           float *__restrict__ hist, int B, int T, int S) {
    const size_t n = (size_t)B * S;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / S);
        const int i = (int)(e - (size_t)b * S);
        hist[(size_t)b * T * S + i] = obs[(size_t)b * T * S + i] + initial[i];
    }
}

// once per decode: empty partial top lists (value -inf, prev-state 0) for both parities
__global__ __launch_bounds__(256) void clear_top_kernel(float *__restrict__ topv, int32_t *__restrict__ topi, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        topv[e] = -INFINITY;
        topi[e] = 0;
    }
}

// kTop largest of the NE values each lane holds (value, tag) across the wave, values descending; ties
// take the lowest tag.  Results are wave-uniform; `emit(r, value, tag)` is called once per rank.
template <int NE, typename Emit>
__device__ __forceinline__ void wave_top(float (&v)[NE], const int (&tag)[NE], Emit emit) {
    unsigned picked = 0;
#pragma unroll
    for (int r = 0; r < kTop; ++r) {
        float lm = -INFINITY;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (!((picked >> e) & 1u)) lm = fmaxf(lm, v[e]);
        const float m = wavered::wave_reduce_f32(lm, wavered::MaxOp());
        int lk = 0x7fffffff, le = 0;
#pragma unroll
        for (int e = NE - 1; e >= 0; --e)
            if (!((picked >> e) & 1u) && tag[e] != 0x7fffffff && v[e] == m) { lk = tag[e]; le = e; }
        const int k = wavered::wave_min_i32(lk);
        if (lk == k && k != 0x7fffffff) picked |= 1u << le;
        emit(r, k == 0x7fffffff ? -INFINITY : m, k == 0x7fffffff ? 0 : k);
    }
}

// once per decode: the kTop largest entries of history row 0 of every item, stored as the partial list of
// state tile 0 (parity 0).  One wave per item; NQ float4 per lane (S <= 256*NQ).  grid = B, block = 64.
template <int NQ>
__global__ __launch_bounds__(64) void top_kernel(const float *__restrict__ hist, float *__restrict__ topv,
                                                 int32_t *__restrict__ topi, int B, int T, int S) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const float *row = hist + (size_t)b * T * S;
    float v[NQ * 4];
    int tag[NQ * 4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = 4 * lane + 256 * q;
        float4 x = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (i < S) x = *reinterpret_cast<const float4 *>(row + i);
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
#pragma unroll
        for (int u = 0; u < 4; ++u) tag[4 * q + u] = i < S ? i + u : 0x7fffffff;
    }
    wave_top<NQ * 4>(v, tag, [&](int r, float m, int k) {
        if (lane == 0) { topv[(size_t)b * kTop + r] = m; topi[(size_t)b * kTop + r] = k; }
    });
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// one rotation step: every lane consumes the list entry held by lane (c + N) % 16 of its next-state
template <int N>
__device__ __forceinline__ void rot_step(float &best, float pt, int poff, int cbytes, const char *tile) {
    float tt;
    int addr;
    if (N == 0) {
        tt = pt;
        addr = poff + cbytes;
    } else {
        asm("v_add_u32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "=v"(addr) : "v"(poff), "v"(cbytes), "i"(N));
        tt = dpp_f<0x120 + (N == 0 ? 1 : N)>(pt);
    }
    best = fmaxf(best, *reinterpret_cast<const float *>(tile + addr) + tt);
}

struct QuadPrefetch {        // everything a group of 4 next-states needs from memory, issued one group ahead
    float seed[kR];
    float2 pf[kLook];
    float first[kLook];      // largest t of each 16-entry block (broadcast load)
    float ob;                // observation of this lane's (item, next-state)
};

// ---------------------------------------------------------------------------------------
// one timestep.  grid = n_bt * n_jt, block = 1024, dynamic LDS = lds_bytes(S, JT).
//
// Per-item top lists travel between timesteps as PARTIAL lists: each state tile leaves the kTop
// largest of the outputs it produced for each of its 16 items (ptop[t & 1][jt][b][r]); the next
// timestep's tiles merge the n_jt partial lists of their items (every member of the global top
// kTop is in the top kTop of its own tile).  No separate selection kernel, no extra launch.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kWaves) void step_pruned_kernel(
    const float *__restrict__ obs, const int32_t *__restrict__ frames, const float *__restrict__ tt,
    const float2 *__restrict__ sorted, const float *__restrict__ ptopv_in, const int32_t *__restrict__ ptopi_in,
    float *__restrict__ ptopv_out, int32_t *__restrict__ ptopi_out, float *__restrict__ hist, int B, int T, int S,
    int t, int SpP, int n_bt, int n_jt, int JT) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *outs = lds + (size_t)kNB * S;                   // [16 items][JT] outputs of this tile
    float *mtopv = outs + (size_t)kNB * JT;                // [16][kTop] merged top values
    int *mtopi = reinterpret_cast<int *>(mtopv + kNB * kTop);
    const int tile_id = blockIdx.x;
    const int bt = tile_id % n_bt, jt = tile_id / n_bt;
    const int b0 = bt * kNB, j0 = jt * JT;
    const int tid = threadIdx.x, lane = tid & 63;
    // skip tiles whose batch items have all ended (t >= batch_frames[b])
    if (!__syncthreads_or(tid < kNB && b0 + tid < B && t < frames[b0 + tid])) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, js = lane >> 4;
    const int b = b0 + c;
    const int bq = b < B ? b : B - 1;
    const bool live = b < B && t < frames[bq];
    const int JTv = S - j0 < JT ? S - j0 : JT;                   // next-states of this tile
    const int nquads = (JTv + 3) / 4;
    const int nb = (S + 15) / 16;

    // wave w merges the partial top lists of item b0 + w: n_jt * kTop candidates, <= 2 per lane
    if (wave < kNB) {
        const int bw = b0 + wave < B ? b0 + wave : B - 1;
        float v[2];
        int tag[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int cand = lane + 64 * e;
            const bool ok = cand < n_jt * kTop;
            const size_t src = ((size_t)(ok ? cand / kTop : 0) * B + bw) * kTop + (ok ? cand % kTop : 0);
            v[e] = ok ? ptopv_in[src] : -INFINITY;
            const int idx = ok ? ptopi_in[src] : 0;
            // tag = prev-state; equal values from different tiles cannot share a prev-state
            tag[e] = ok ? idx : 0x7fffffff;
        }
        wave_top<2>(v, tag, [&](int r, float m, int k) {
            if (lane == 0) { mtopv[wave * kTop + r] = m; mtopi[wave * kTop + r] = k; }
        });
    }

    // stage the 16 posterior rows as [prev-state][16 items]: lanes = 16 rows x 4 float4 columns
    {
        const int n4 = kNB * (S / 4);
        for (int e0 = tid; e0 < n4; e0 += 4 * 64 * kWaves) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 64 * kWaves;
                const int bb = e & (kNB - 1), i4 = e / kNB;
                const int brow = b0 + bb < B ? b0 + bb : B - 1;
                v[u] = e < n4 ? *reinterpret_cast<const float4 *>(hist + ((size_t)brow * T + (t - 1)) * S + 4 * i4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 64 * kWaves;
                if (e < n4) {
                    const int bb = e & (kNB - 1), i4 = e / kNB;
                    float *d = lds + (4 * i4) * kNB + bb;
                    d[0] = v[u].x; d[kNB] = v[u].y; d[2 * kNB] = v[u].z; d[3 * kNB] = v[u].w;
                }
            }
        }
    }
    __syncthreads();

    float tv[kR];
    int ti[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) { tv[r] = mtopv[c * kTop + r]; ti[r] = mtopi[c * kTop + r]; }
    const float thr = mtopv[c * kTop + kR];

    auto issue = [&](QuadPrefetch &pre, int q) {
        const int jj = 4 * q + js;
        const int jr = jj < JTv ? j0 + jj : j0;
#pragma unroll
        for (int r = 0; r < kR; ++r) pre.seed[r] = tt[(size_t)ti[r] * S + jr];   // trans[jr][i_r]
        const float2 *row = sorted + (size_t)jr * SpP + c;
#pragma unroll
        for (int u = 0; u < kLook; ++u) { pre.pf[u] = row[16 * u]; pre.first[u] = row[16 * u - c].x; }
        pre.ob = obs[((size_t)bq * T + t) * S + jr];
    };
    QuadPrefetch cur, nxt;
    if (wave < nquads) issue(cur, wave);

    const char *ptile = reinterpret_cast<const char *>(lds);
    const int cbytes = 4 * c;
    for (int q = wave; q < nquads; q += kWaves) {
        if (q + kWaves < nquads) issue(nxt, q + kWaves);
        const int jj = 4 * q + js;
        const bool jv = jj < JTv;
        const int jr = jv ? j0 + jj : j0;
        float best = -INFINITY;
#pragma unroll
        for (int r = 0; r < kR; ++r) best = fmaxf(best, tv[r] + cur.seed[r]);
        const float2 *row = sorted + (size_t)jr * SpP + c;
        for (int kb = 0; kb < nb; kb += kLook) {
#pragma unroll
            for (int u = 0; u < kLook; u += 2) {
                // two 16-entry blocks per test: stop once no lane's bound t_first + thr exceeds its best
                if (kb + u >= nb || !__any(jv && cur.first[u] + thr > best)) goto done;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float pt = cur.pf[u + h].x;
                    const int pi = __builtin_bit_cast(int, cur.pf[u + h].y);
                    // rows are padded with 16*kLook (-inf) entries: the prefetch never leaves the row
                    cur.pf[u + h] = row[16 * (kb + u + h + kLook)];
                    cur.first[u + h] = row[16 * (kb + u + h + kLook) - c].x;
                    rot_step<0>(best, pt, pi, cbytes, ptile); rot_step<1>(best, pt, pi, cbytes, ptile);
                    rot_step<2>(best, pt, pi, cbytes, ptile); rot_step<3>(best, pt, pi, cbytes, ptile);
                    rot_step<4>(best, pt, pi, cbytes, ptile); rot_step<5>(best, pt, pi, cbytes, ptile);
                    rot_step<6>(best, pt, pi, cbytes, ptile); rot_step<7>(best, pt, pi, cbytes, ptile);
                    rot_step<8>(best, pt, pi, cbytes, ptile); rot_step<9>(best, pt, pi, cbytes, ptile);
                    rot_step<10>(best, pt, pi, cbytes, ptile); rot_step<11>(best, pt, pi, cbytes, ptile);
                    rot_step<12>(best, pt, pi, cbytes, ptile); rot_step<13>(best, pt, pi, cbytes, ptile);
                    rot_step<14>(best, pt, pi, cbytes, ptile); rot_step<15>(best, pt, pi, cbytes, ptile);
                }
            }
        }
    done:
        {
            const float o = cur.ob + best;                                    // post'[j] = obs[t,j] + max
            if (jv && live) hist[((size_t)b * T + t) * S + jr] = o;
            if (jv) outs[c * JT + jj] = o;
        }
        cur = nxt;
    }
    __syncthreads();

    // partial top list of this tile for each of its items: wave w scans item w's JTv outputs
    if (wave < kNB) {
        constexpr int NE = 4;                 // JT <= 256 outputs per item -> 4 per lane
        float v[NE];
        int tag[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int jj = lane + 64 * e;
            const bool ok = jj < JTv;
            v[e] = ok ? outs[wave * JT + jj] : -INFINITY;
            tag[e] = ok ? j0 + jj : 0x7fffffff;
        }
        const int bw = b0 + wave;
        wave_top<NE>(v, tag, [&](int r, float m, int k) {
            if (lane == 0 && bw < B) {
                ptopv_out[((size_t)jt * B + bw) * kTop + r] = m;
                ptopi_out[((size_t)jt * B + bw) * kTop + r] = k;
            }
        });
    }
}

}  // namespace pruned
