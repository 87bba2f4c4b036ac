// torbi_hip.hip -- MI355X (gfx950 / CDNA4) batched Viterbi decoder behind a plain C ABI.
//
// Written from scratch for wave64 / LDS / 256-CU CDNA4; not derived from the reference's
// CUDA kernels.  The reference functions each piece replaces are cited by file:line
// (paths relative to /root/reference/torbi/csrc/).
//
//   forward recurrence   viterbi.cpp:65-108 (oracle semantics) / cuda/viterbi.cu:48-130
//   final argmax + fill  viterbi.cpp:218-221                   / cuda/viterbi.cu:347-350
//   backtrace            viterbi.cpp:140-160                   / cuda/viterbi.cu:150-176
//   orchestration        viterbi.cpp:182-234                   / cuda/viterbi.cu:309-362
//
// The forward recurrence is a (max,+) matrix product per timestep,
//     post'[b,j] = obs[b,t,j] + max_i ( post[b,i] + trans[j,i] ),   bp[b,t,j] = first argmax_i
// i.e. a GEMM-shaped contraction over i with M = batch, N = K = states, in the (max,+)
// semiring: VALU work (MFMA cannot evaluate it), 2 fp32 roundings per cell in a fixed order.
// See DESIGN.md for the roofline analysis and kernel inventory.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <stdlib.h>

#include "torbi_hip.h"
#include "dense_forward.hpp"
#include "lazy_backtrace.hpp"
#include "uniform_decode.hpp"
#include "pruned_forward.hpp"
#include "resident_forward.hpp"
#include "small_batch_forward.hpp"
#include "held_matrix_forward.hpp"
#include "small_states.hpp"
#include "band_forward.hpp"
#include "band_tile_forward.hpp"

namespace {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// (value, index) comparator of the reference CPU scan (viterbi.cpp:94-100): a candidate
// replaces the incumbent only if strictly greater; among equal values the lower index wins.
__device__ __forceinline__ void take_better(float &v, int &i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__device__ __forceinline__ void wave_argmax(float &v, int &i) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const float ov = __shfl_down(v, off, kWave);
        const int oi = __shfl_down(i, off, kWave);
        take_better(v, i, ov, oi);
    }
}

// ---------------------------------------------------------------------------------------
// t = 0 : post[b,i] = obs[b,0,i] + initial[i]                       (viterbi.cpp:72-76)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_posterior_kernel(
    const float *__restrict__ obs, const float *__restrict__ initial, float *__restrict__ post0,
    int B, int T, int S) {
    const size_t n = (size_t)B * S;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / S);
        const int i = (int)(e - (size_t)b * S);
        post0[e] = obs[(size_t)b * T * S + i] + initial[i];
    }
}

// ---------------------------------------------------------------------------------------
// One timestep, small-batch form: grid = (state tiles, batch items); each wave owns 4
// next-states, its 64 lanes stride the prev-state axis (coalesced transition-row reads,
// posterior row staged once in LDS), then a wave (value,index) reduction that keeps the
// lowest index among equal maxima.
// ---------------------------------------------------------------------------------------
constexpr int kRowsPerWave = 4;
constexpr int kRowsPerBlock = 16;

__global__ __launch_bounds__(256) void step_rows_kernel(
    const float *__restrict__ obs, const int32_t *__restrict__ frames,
    const float *__restrict__ trans, const float *__restrict__ pcur, float *__restrict__ pnext,
    int32_t *__restrict__ trellis, int B, int T, int S, int t, int b0) {
    extern __shared__ __attribute__((aligned(16))) float pl[];
    const int b = b0 + blockIdx.y;            // (gridDim.y holds at most 65535 items: larger batches take several launches)
    if (t >= frames[b]) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < S; i += 256) pl[i] = pcur[(size_t)b * S + i];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int jb = blockIdx.x * kRowsPerBlock + wave * kRowsPerWave;
    if (jb >= S) return;
    const float *rows[kRowsPerWave];
    float best[kRowsPerWave];
    int arg[kRowsPerWave];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int j = min(jb + r, S - 1);
        rows[r] = trans + (size_t)j * S;
        best[r] = -INFINITY;
        arg[r] = 0x7fffffff;   // lanes that saw no candidate must lose index ties
    }
    for (int i = lane; i < S; i += kWave) {
        const float p = pl[i];
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            const float c = p + rows[r][i];
            if (c > best[r]) { best[r] = c; arg[r] = i; }
            else if (arg[r] == 0x7fffffff) { arg[r] = i; best[r] = c; }  // first candidate, incl. -inf
        }
    }
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) wave_argmax(best[r], arg[r]);
    if (lane == 0) {
        const size_t row = ((size_t)b * T + t) * S;
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            const int j = jb + r;
            if (j < S) {
                trellis[row + j] = arg[r];
                pnext[(size_t)b * S + j] = obs[row + j] + best[r];
            }
        }
    }
}

// Small-batch timestep, vector form (S % 4 == 0, 16-byte aligned rows): one next-state per wave,
// 4 per block, so a batch-1 step spreads over S/4 workgroups; each lane owns 4 consecutive
// prev-states per 256-wide stripe (ascending per lane, like the reference scan) and loads the
// transition row as float4.
__global__ __launch_bounds__(256) void step_rows4_kernel(
    const float *__restrict__ obs, const int32_t *__restrict__ frames,
    const float *__restrict__ trans, const float *__restrict__ pcur, float *__restrict__ pnext,
    int32_t *__restrict__ trellis, int B, int T, int S, int t, int b0) {
    extern __shared__ __attribute__((aligned(16))) float pl[];
    const int b = b0 + blockIdx.y;
    if (t >= frames[b]) return;
    const int tid = threadIdx.x;
    {
        const float4 *src = reinterpret_cast<const float4 *>(pcur + (size_t)b * S);
        float4 *dst = reinterpret_cast<float4 *>(pl);
        for (int i = tid; i < S / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x * 4 + wave;
    if (j >= S) return;
    const float *row = trans + (size_t)j * S;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int i = 4 * lane; i < S; i += 256) {
        const float4 q = *reinterpret_cast<const float4 *>(row + i);
        const float4 p = *reinterpret_cast<const float4 *>(pl + i);
        const float c[4] = {p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c[u] > best) { best = c[u]; arg = i + u; }
            else if (arg == 0x7fffffff) { arg = i + u; best = c[u]; }
        }
    }
    wave_argmax(best, arg);
    if (lane == 0) {
        const size_t e = ((size_t)b * T + t) * S + j;
        trellis[e] = arg;
        pnext[(size_t)b * S + j] = obs[e] + best;
    }
}

// ---------------------------------------------------------------------------------------
// Final state + tail fill + backtrace, one wave per batch item.
//   final = first argmax of the item's last posterior row           (viterbi.cpp:218)
//   out[b, t] = final for t >= frames-1                               (viterbi.cpp:219-221)
//   for t = frames-1 .. 1: idx = bp[b,t,idx]; out[b,t-1] = idx        (viterbi.cpp:153-157)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void finalize_kernel(
    const float *__restrict__ post0, const float *__restrict__ post1,
    const int32_t *__restrict__ frames, const int32_t *__restrict__ trellis,
    int32_t *__restrict__ out, int B, int T, int S) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    int f = frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    const float *post = (((f - 1) & 1) ? post1 : post0) + (size_t)b * S;

    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int i = lane; i < S; i += kWave) {
        const float v = post[i];
        if (v > best) { best = v; arg = i; }
        else if (arg == 0x7fffffff) { arg = i; best = v; }
    }
    wave_argmax(best, arg);
    const int fin = __shfl(arg, 0, kWave);

    int32_t *o = out + (size_t)b * T;
    for (int tt = f - 1 + lane; tt < T; tt += kWave) o[tt] = fin;
    if (lane == 0) {
        const int32_t *tr = trellis + (size_t)b * T * S;
        int idx = fin;
        for (int tt = f - 1; tt >= 1; --tt) {
            idx = tr[(size_t)tt * S + idx];
            o[tt - 1] = idx;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The chase in parallel (a handful of long sequences: one lane following 500 dependent loads at ~265 ns each is 0.13 ms,
// an eighth of a whole batch-of-one decode).  A backpointer row is a function S -> S and the path is their composition
// applied to the final state, so the timesteps are cut into chunks of kChaseChunk:
//   chase_maps_kernel     every chunk at once, every state at once: where does a path that is in state s at the chunk's
//                         upper end stand at its lower end (kChaseChunk dependent lookups, all rows L1/L2-resident);
//   chase_entries_kernel  one wave per item: final argmax + tail fill as in finalize_kernel, the top partial chunk
//                         directly, then chunk map after chunk map: the state at every chunk boundary;
//   chase_chunks_kernel   every chunk at once: the path inside the chunk from its upper boundary state.
// Three short launches instead of T dependent loads: 1 x 500 x 1440 in ~40 us instead of 132 us.  Same indices.
// ---------------------------------------------------------------------------------------
constexpr int kChaseChunk = 32;
constexpr int kChaseMinSteps = 128;        // below this the plain chase is as fast
inline int chase_chunks(int T) { return (T - 1 + kChaseChunk - 1) / kChaseChunk; }

__device__ __forceinline__ int clamped_frames(const int32_t *frames, int b, int T) {
    const int f = frames[b];
    return f < 1 ? 1 : (f > T ? T : f);
}

// grid = (chunks, B, ceil(S / 256)), block = 256: maps[b][k][s] = state at time k * C of the path that is in s at time (k + 1) * C
__global__ __launch_bounds__(256) void chase_maps_kernel(const int32_t *__restrict__ frames, const int32_t *__restrict__ trellis,
                                                         int32_t *__restrict__ maps, int B, int T, int S, int chunks) {
    const int k = blockIdx.x, b = blockIdx.y;
    const int s = blockIdx.z * 256 + threadIdx.x;
    const int last = clamped_frames(frames, b, T) - 1;          // the path ends at time `last`
    const int hi = (k + 1) * kChaseChunk;
    if (hi > last || s >= S) return;                            // only chunks the path crosses completely
    const int32_t *tr = trellis + (size_t)b * T * S;
    int idx = s;
    for (int t = hi; t > k * kChaseChunk; --t) idx = tr[(size_t)t * S + idx];
    maps[((size_t)b * chunks + k) * S + s] = idx;
}

// grid = B, block = 64
__global__ __launch_bounds__(64) void chase_entries_kernel(const float *__restrict__ post0, const float *__restrict__ post1,
                                                           const int32_t *__restrict__ frames, const int32_t *__restrict__ trellis,
                                                           const int32_t *__restrict__ maps, int32_t *__restrict__ entries,
                                                           int32_t *__restrict__ out, int B, int T, int S, int chunks) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int f = clamped_frames(frames, b, T);
    const float *post = (((f - 1) & 1) ? post1 : post0) + (size_t)b * S;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int i = lane; i < S; i += kWave) {
        const float v = post[i];
        if (v > best) { best = v; arg = i; }
        else if (arg == 0x7fffffff) { arg = i; best = v; }
    }
    wave_argmax(best, arg);
    const int fin = __shfl(arg, 0, kWave);
    int32_t *o = out + (size_t)b * T;
    for (int tt = f - 1 + lane; tt < T; tt += kWave) o[tt] = fin;
    if (lane == 0) {
        const int32_t *tr = trellis + (size_t)b * T * S;
        const int last = f - 1;
        const int k0 = last / kChaseChunk;                      // complete chunks below the path's end: 0 .. k0 - 1
        int idx = fin;
        for (int tt = last; tt > k0 * kChaseChunk; --tt) {      // the partial chunk at the top, directly
            idx = tr[(size_t)tt * S + idx];
            o[tt - 1] = idx;
        }
        for (int k = k0 - 1; k >= 0; --k) {                     // idx = state at time (k + 1) * C
            entries[(size_t)b * chunks + k] = idx;
            idx = maps[((size_t)b * chunks + k) * S + idx];
            o[k * kChaseChunk] = idx;
        }
    }
}

// grid = (chunks, B), block = 64: the path inside chunk k from its upper boundary state
__global__ __launch_bounds__(64) void chase_chunks_kernel(const int32_t *__restrict__ frames, const int32_t *__restrict__ trellis,
                                                          const int32_t *__restrict__ entries, int32_t *__restrict__ out,
                                                          int B, int T, int S, int chunks) {
    const int k = blockIdx.x, b = blockIdx.y;
    const int last = clamped_frames(frames, b, T) - 1;
    if ((k + 1) * kChaseChunk > last || threadIdx.x != 0) return;
    const int32_t *tr = trellis + (size_t)b * T * S;
    int32_t *o = out + (size_t)b * T;
    int idx = entries[(size_t)b * chunks + k];
    for (int t = (k + 1) * kChaseChunk; t > k * kChaseChunk + 1; --t) {
        idx = tr[(size_t)t * S + idx];
        o[t - 1] = idx;
    }
}

// final posterior rows of the last decode, whatever path it took (route record): the generic path keeps them in its
// ping-pong buffers, every value-only path as row frames-1 of the history at the start of the workspace
__global__ __launch_bounds__(256) void gather_final_kernel(const int32_t *__restrict__ route, const float *__restrict__ hist,
                                                           const float *__restrict__ post0, const float *__restrict__ post1,
                                                           const int32_t *__restrict__ frames, float *__restrict__ dst,
                                                           int B, int T, int S) {
    const bool generic = *route == 0 || *route == 6 || *route == 7;       // trellis kernels: per-timestep, held-matrix, one wavefront
    const size_t n = (size_t)B * S;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / S);
        const int i = (int)(e - (size_t)b * S);
        int f = frames[b];
        f = f < 1 ? 1 : (f > T ? T : f);
        dst[e] = generic ? (((f - 1) & 1) ? post1 : post0)[e] : hist[((size_t)b * T + (f - 1)) * S + i];
    }
}

// scan statistics of the last decode by its route record: time-resident forms (and the held kernel's give-ups), else zeros
__global__ __launch_bounds__(128) void gather_stats_kernel(const int32_t *__restrict__ route, const unsigned *__restrict__ resident_stats,
                                                           const unsigned *__restrict__ held_control, unsigned *__restrict__ dst) {
    const int r = *route;
    const unsigned *src = (r == 3 || r == 5 || r == 8) ? resident_stats : nullptr;      // (band launches count their give-ups in [127] too)
    unsigned v = src ? src[threadIdx.x] : 0u;
    // held-matrix launch: [127] = workgroups that gave up waiting (the decode was then repaired; 0 on any sane run)
    if (r == 6 && held_control && threadIdx.x == 127) v = held_control[1];
    // dense launches: [120], [121] = shader-clock / 100 MHz wall-clock ticks of workgroup 0 of the last timestep's launch (the
    // time-resident and band kernels leave theirs in the statistics themselves)
    if (r == 1 && (threadIdx.x == 120 || threadIdx.x == 121)) v = (unsigned)route[2 + (threadIdx.x - 120)];
    dst[threadIdx.x] = v;
}

// x <- log(exp(x) + tiny), the epsilon clamp of from_probabilities (torbi/core.py:193-197)
__global__ __launch_bounds__(256) void epsilon_clamp_kernel(float *__restrict__ x, uint64_t count) {
    const float tiny = 1.17549435e-38f;   // torch.finfo(torch.float32).tiny
    const uint64_t n4 = count / 4;
    float4 *x4 = reinterpret_cast<float4 *>(x);
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4;
         e += (uint64_t)gridDim.x * blockDim.x) {
        float4 v = x4[e];
        v.x = logf(expf(v.x) + tiny);
        v.y = logf(expf(v.y) + tiny);
        v.z = logf(expf(v.z) + tiny);
        v.w = logf(expf(v.w) + tiny);
        x4[e] = v;
    }
    for (uint64_t e = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count;
         e += (uint64_t)gridDim.x * blockDim.x)
        x[e] = logf(expf(x[e]) + tiny);
}

// y = log(exp(log(p)) + tiny): the log() of probability inputs (torbi/core.py:189-191) and the epsilon clamp behind it
// (core.py:193-197) in one pass, out of place like upstream's torch.log -- or in place (y == p: every element is read and
// written by one thread; the many-file driver's staged batches)
__global__ __launch_bounds__(256) void log_epsilon_clamp_kernel(const float *p, float *y, uint64_t count) {
    const float tiny = 1.17549435e-38f;
    const uint64_t n4 = count / 4;
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4;
         e += (uint64_t)gridDim.x * blockDim.x) {
        float4 v = p4[e];
        v.x = logf(expf(logf(v.x)) + tiny);
        v.y = logf(expf(logf(v.y)) + tiny);
        v.z = logf(expf(logf(v.z)) + tiny);
        v.w = logf(expf(logf(v.w)) + tiny);
        y4[e] = v;
    }
    for (uint64_t e = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count;
         e += (uint64_t)gridDim.x * blockDim.x)
        y[e] = logf(expf(logf(p[e])) + tiny);
}

// deterministic synthetic scores, same function as torbi_amd/synth.py::scores
__global__ __launch_bounds__(256) void fill_synthetic_kernel(float *__restrict__ dst,
                                                             uint64_t count, uint64_t start,
                                                             uint64_t stream_key) {
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count;
         e += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = (start + e) + stream_key;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        const uint32_t u = (uint32_t)(z >> 40);
        dst[e] = 0.0f - (float)u * 0x1p-20f;
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

// Compute units of a device (tiling plans are functions of (B, S, CUs)); queried once per device.
constexpr int kMaxDevices = 64;
std::atomic<int> g_cus[kMaxDevices];
inline int cu_count(int device) {
    if (device < 0 || device >= kMaxDevices) return 256;
    int v = g_cus[device].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
    g_cus[device].store(n, std::memory_order_relaxed);
    return n;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per (kernel, device): set once, and again only for a larger request
struct LdsGrant { const void *fn; int device; size_t bytes; };
std::mutex g_lds_mutex;
std::vector<LdsGrant> g_lds_grants;
inline hipError_t ensure_dynamic_lds(const void *fn, size_t bytes) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> hold(g_lds_mutex);
    for (auto &g : g_lds_grants)
        if (g.fn == fn && g.device == device) {
            if (g.bytes >= bytes) return hipSuccess;
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) g.bytes = bytes;
            return e;
        }
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) g_lds_grants.push_back(LdsGrant{fn, device, bytes});
    return e;
}

// Decodes of this library that may still be running on a device, by stream: the end of every decode is marked with an
// event on its stream.  Read by AUTO before it picks the held-matrix kernel, whose workgroups must all be resident at
// once (held_matrix_forward.hpp): beside a launch on ANOTHER stream that holds the compute units -- a time-resident
// launch group takes all of them for tens of milliseconds, a second held launch can leave both partly resident -- its
// workgroups would spin until their time budget runs out and the slow repair kernel would decode the call.  With another
// stream busy a handful of sequences take the per-timestep kernels instead, which simply queue behind the other work.
// (Work this library did not launch is invisible here; the time budget and the repair kernel cover it.)
struct StreamMark { hipStream_t stream; hipEvent_t done; };
std::mutex g_marks_mutex;
std::vector<StreamMark> g_marks[kMaxDevices];
constexpr size_t kMaxMarks = 64;

// `s` is being captured into a HIP graph: nothing may be asked of an event (hipEventQuery invalidates a capture), and an
// event recorded now becomes a node of the graph that no later query may name
inline bool capturing(hipStream_t s) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &status) != hipSuccess) { (void)hipGetLastError(); return false; }
    return status != hipStreamCaptureStatusNone;
}

inline void mark_decode_end(int device, hipStream_t s) {
    if (device < 0 || device >= kMaxDevices || capturing(s)) return;
    std::lock_guard<std::mutex> hold(g_marks_mutex);
    auto &marks = g_marks[device];
    StreamMark *slot = nullptr;
    for (auto &m : marks)
        if (m.stream == s) { slot = &m; break; }
    if (!slot && marks.size() >= kMaxMarks) {           // full: take over the mark of a stream that has gone quiet
        for (auto &m : marks)
            if (hipEventQuery(m.done) == hipSuccess) { slot = &m; break; }
        if (!slot) return;
        slot->stream = s;
    }
    if (!slot) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
        marks.push_back(StreamMark{s, ev});
        slot = &marks.back();
    }
    if (hipEventRecord(slot->done, s) != hipSuccess) (void)hipGetLastError();
}

inline bool other_streams_busy(int device, hipStream_t s) {
    // (while capturing: what the replays will run beside is unknown anyway; the held kernel's bounded waits and its
    // repair launch cover a busy device)
    if (device < 0 || device >= kMaxDevices || capturing(s)) return false;
    std::lock_guard<std::mutex> hold(g_marks_mutex);
    for (auto &m : g_marks[device])
        if (m.stream != s) {
            const hipError_t q = hipEventQuery(m.done);
            if (q == hipErrorNotReady) return true;
            if (q != hipSuccess) (void)hipGetLastError();
        }
    return false;
}

// name of the forward kernel the calling thread's most recent decode launched, spelled as rocprofv3 prints it
// (torbi_hip_last_forward_kernel: bench.py matches it against the committed counter summaries)
thread_local char g_last_kernel[160] = "";
#include <stdio.h>
#define TORBI_NOTE_KERNEL(...) snprintf(g_last_kernel, sizeof(g_last_kernel), __VA_ARGS__)

// Which forward recurrence runs.  GENERIC and HELD materialise the int32 trellis like the reference does; the others
// keep the posterior history and recompute backpointers along the decoded path (lazy_backtrace.hpp).  (2 was the
// per-timestep tile kernel of the pruned recurrence, removed in round 4: no route has the number any more.)
enum Route { ROUTE_GENERIC = 0, ROUTE_DENSE = 1, ROUTE_PRUNED = 2, ROUTE_RESIDENT = 3, ROUTE_ROWS = 4, ROUTE_CLUSTER = 5,
             ROUTE_HELD = 6, ROUTE_SMALL = 7, ROUTE_BAND = 8 };

inline bool use_dense(int B, int S) { return B >= 32 && S >= 64; }

// process-wide default path: torbi_hip_set_forward_path / TORBI_HIP_FORWARD=dense|pruned|resident (read once).
// A call that carries a path in its flags ignores it.
std::atomic<int> g_forward_path{-1};
inline int default_path() {
    int v = g_forward_path.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TORBI_HIP_FORWARD");
        v = !e ? TORBI_HIP_FORWARD_AUTO
               : e[0] == 'd' ? TORBI_HIP_FORWARD_DENSE
               : e[0] == 'p' ? TORBI_HIP_FORWARD_PRUNED
               : e[0] == 'r' ? TORBI_HIP_FORWARD_RESIDENT
               : e[0] == 'c' ? TORBI_HIP_FORWARD_CLUSTER
               : e[0] == 'h' ? TORBI_HIP_FORWARD_HELD
               : e[0] == 'b' ? TORBI_HIP_FORWARD_BAND : TORBI_HIP_FORWARD_AUTO;
        g_forward_path.store(v, std::memory_order_relaxed);
    }
    return v;
}
// path carried by the flags of a call ((path + 1) << 4), else the process default
inline int requested_path(unsigned flags) {
    const unsigned f = (flags >> 4) & 7u;
    return f ? (int)f - 1 : default_path();
}
constexpr unsigned kKnownFlags = TORBI_HIP_REUSE_TRANSITION | TORBI_HIP_COLLECT_STATS | (7u << 4) | TORBI_HIP_SHORTEST_FIRST |
                                 TORBI_HIP_FEW_SEEDS | TORBI_HIP_MANY_SEEDS;
inline bool flags_ok(unsigned flags) {
    return !(flags & ~kKnownFlags) && ((flags >> 4) & 7u) <= (unsigned)TORBI_HIP_FORWARD_BAND + 1u;
}

constexpr int kMaxGroupTiles = 16384;     // 16-item tiles one launch group may hold (262 144 items)
// tiles of a batch in the time-resident kernel: 16 items each, 8 above 2048 states
inline int tiles_of(int B, int S) { const int ni = resident::tile_items(S); return (B + ni - 1) / ni; }
inline bool resident_fits(int S, int tiles) { return resident::supported(S) && tiles <= kMaxGroupTiles; }

// members per cluster for a launch of `tiles` 16-item tiles: 1 (every workgroup owns a whole tile) once the tiles give
// at least half the compute units a workgroup, else as many as the compute units allow (<= kMaxR, <= one row group each)
inline int cluster_members(int tiles, int S, int cus) {
    if (tiles < 1 || 2 * tiles > cus) return 1;
    int R = cus / tiles;
    const int nrg = (S + resident::pass_rows(S) - 1) / resident::pass_rows(S);
    R = std::min(R, std::min(resident::kMaxR, nrg));
    return R < 2 ? 1 : R;
}

// AUTO takes the held-matrix kernel up to this many items (TORBI_HIP_HELD_ITEMS overrides; 0 = never).  tools/held_probe.py,
// ms per decode against the best per-timestep kernel (profiles/r03_held_probe.txt): 500 frames x 1440 states 1 item
// 1.10 / 2.41, 2 items 1.38 / 2.52, 3 items 2.16 / 2.69, 4 items 2.83 / 2.85, 8 items 5.44 / 4.31; 300 frames x 4096 states
// 1 item 1.28 / 3.87, 2 items 1.86 / 4.20, 4 items 3.79 / 5.95, 8 items 7.41 / 9.34.  A timestep of the kernel costs one
// hand-off (~1.5 us) for up to two sequences and ~1.3 us of instruction issue for every further one; a launch of the
// per-timestep kernels 4.4 us plus ~0.3 us per item.
inline bool held_auto(int B, int S) {
    static const int limit = [] {
        const char *e = getenv("TORBI_HIP_HELD_ITEMS");
        return e ? atoi(e) : -1;
    }();
    return B <= (limit >= 0 ? limit : S > held::kSmallS ? 8 : 3);
}

// route of ONE batch.  AUTO: the time-resident kernel -- whole tiles per workgroup when the batch alone gives at least
// half the compute units a workgroup, tiles split over clusters of workgroups for smaller batches of >= 17 items --
// else the sorted-row scan / held-matrix kernel for a handful of sequences, else the dense (max,+) GEMM, else generic.
inline bool small_block_auto(int B, int S, int cus) {
    // (value-only since round 5: ahead of the time-resident kernels at every batch size up to 192 states -- 8192 x 200 x 128
    // 2.0 against 4.0 ms, 8192 x 100 x 192 2.8 against 3.0 -- and up to ~3000 sequences at 256: 2048 x 200 x 256 1.6 against
    // 2.3 ms, 4096 x 200 x 256 3.2 against 2.9.  TORBI_HIP_SMALL_BLOCK_LIMIT: cells per compute unit, in units of 65536)
    static const long long limit = [] {
        const char *e = getenv("TORBI_HIP_SMALL_BLOCK_LIMIT");
        return e ? atoll(e) : 12ll;
    }();
    return small::block_supported(S) && (S <= 192 || (long long)B * S * S <= (limit << 16) * cus);
}
inline Route route_for(int path, int B, int S, int cus, bool allow_held = true) {
    if (path == TORBI_HIP_FORWARD_BAND) path = TORBI_HIP_FORWARD_AUTO;     // (a band is known to torbi_hip_viterbi_decode_banded only)
    const bool fits = resident_fits(S, tiles_of(B, S));
    // up to 64 states a wavefront decodes a sequence on its own, time loop and backtrace in one launch (small_states.hpp)
    // ... up to 256 a workgroup does (the matrix in the registers of one compute unit), while the batch is not so large that
    // the time-resident forms' pruning overtakes it: tools/small_states_probe.py, 500 frames, small / resident ms --
    // 4096 x 128: 3.7 / 4.2, 512 x 256: 2.3 / 3.5 (cluster), 1024 x 256: 4.2 / 3.4, 4096 x 256: 19.2 / 6.0 -- B S^2 <= 3 * 2^16 per
    // compute unit (768 items at 256 states, 3072 at 128: the kernel runs one workgroup, or four, per unit at a time)
    if (path == TORBI_HIP_FORWARD_AUTO && (small::supported(S) || small_block_auto(B, S, cus))) return ROUTE_SMALL;
    if (path == TORBI_HIP_FORWARD_RESIDENT && fits) return ROUTE_RESIDENT;
    if (path == TORBI_HIP_FORWARD_CLUSTER && fits) return cluster_members(tiles_of(B, S), S, cus) > 1 ? ROUTE_CLUSTER : ROUTE_RESIDENT;
    if (path == TORBI_HIP_FORWARD_AUTO && fits && 2 * tiles_of(B, S) > cus) return ROUTE_RESIDENT;
    // (one batch, AUTO: clusters for every batch of more than 16 items -- with one seed per item and one polling wave per
    // workgroup the cluster form is at least as fast as the per-timestep kernel from 17 items on: 12.8 against 14.5 us per
    // timestep at 17 items, 15.6 against 20.2 at 512, 20.4 against 34.9 at 768 (profiles/r03_cluster_sweep_one_poller.txt);
    // 128 x 2000 x 4096 on 8-item tiles 54.2 against 55.9 ms)
    if (path == TORBI_HIP_FORWARD_AUTO && fits && B > 16 && cluster_members(tiles_of(B, S), S, cus) > 1)
        return ROUTE_CLUSTER;
    // PRUNED named for more than 16 items: the pruned recurrence in its time-resident form (its per-timestep tile kernel was
    // removed in round 4; up to 16 items PRUNED means the sorted-row scan below)
    if (path == TORBI_HIP_FORWARD_PRUNED && fits && B > 16)
        return cluster_members(tiles_of(B, S), S, cus) > 1 ? ROUTE_CLUSTER : ROUTE_RESIDENT;
    // a handful of sequences: the whole time loop in one launch, the matrix held in registers across the chip
    // (held_matrix_forward.hpp) -- AUTO up to three items, eight above 2048 states (held_auto); any B <= 16 when named
    if (held::supported(B, S, cus) &&
        (path == TORBI_HIP_FORWARD_HELD || (path == TORBI_HIP_FORWARD_AUTO && allow_held && held_auto(B, S))))
        return ROUTE_HELD;
    // up to 16 items the sorted-row scan's time grows with items x states, the cluster form's (one tile, sixteen members)
    // does not: 12 x 1440 rows 5.1 / cluster 5.3 ms per 500 frames, 16 x 1440 6.1 / 5.3, 12 x 2048 4.6 / 4.2 per 300,
    // 12 x 4096 8.7 / 4.8 per 200, 16 x 4096 10.9 / 4.8 -- but 16 x 512 2.4 / 3.6 (tools/few_items_probe.py)
    if (path == TORBI_HIP_FORWARD_AUTO && fits && (long long)B * S > 12ll * 1440 &&
        cluster_members(tiles_of(B, S), S, cus) > 1)
        return ROUTE_CLUSTER;
    if ((path == TORBI_HIP_FORWARD_PRUNED && rowscan::supported(B, S)) ||
        (path != TORBI_HIP_FORWARD_DENSE && rowscan::profitable(B, S)))
        return ROUTE_ROWS;
    // a named path that does not cover the shape falls back as AUTO would (DENSE named below 32 items: the generic kernels)
    if (path != TORBI_HIP_FORWARD_AUTO && path != TORBI_HIP_FORWARD_DENSE)
        return route_for(TORBI_HIP_FORWARD_AUTO, B, S, cus, allow_held);
    return use_dense(B, S) ? ROUTE_DENSE : ROUTE_GENERIC;
}

// ---- workspace layouts: the (B,T,S) history / trellis first, the per-transition preparation behind it ---------
inline size_t history_bytes(int B, int T, int S) { return align_up(sizeof(float) * (size_t)B * T * S, 256); }

struct Workspace {
    float *post[2];     // (B,S) ping-pong posterior rows
    int32_t *trellis;   // (B,T,S) backpointers; rows t >= 1 of valid frames are written
    held::u64 *xchg;    // [2][B][S] {posterior, timestep} words of the held-matrix kernel (B <= 16, S <= 2048), else null
    unsigned *control;  // [64] its control words ([1]: workgroups that gave up waiting)
    int32_t *maps;      // [B][chunks][S] chunk maps of the parallel chase (B <= 16, S <= 4096, T >= 129), else null
    int32_t *entries;   // [B][chunks]    the path's state at every chunk boundary
    int32_t *arrive;    // [B][8] ends of the backtrace's speculative segments (65 .. 256 states, B <= 1024), else null
    size_t bytes;
};

inline Workspace carve(void *base, int B, int T, int S) {
    Workspace w;
    char *p = static_cast<char *>(base);
    const size_t post_bytes = align_up(sizeof(float) * (size_t)B * S, 256);
    w.trellis = reinterpret_cast<int32_t *>(p);
    w.post[0] = reinterpret_cast<float *>(p + history_bytes(B, T, S));
    w.post[1] = reinterpret_cast<float *>(p + history_bytes(B, T, S) + post_bytes);
    w.bytes = history_bytes(B, T, S) + 2 * post_bytes;
    w.xchg = nullptr;
    w.control = nullptr;
    if (B <= held::kMaxB && S <= held::kMaxS) {
        w.control = reinterpret_cast<unsigned *>(p + w.bytes);
        w.xchg = reinterpret_cast<held::u64 *>(p + w.bytes + 256);
        w.bytes += 256 + align_up(held::exchange_bytes(B, S), 256);
    }
    w.maps = nullptr;
    w.entries = nullptr;
    if (B <= held::kMaxB && S <= held::kMaxS && T - 1 >= kChaseMinSteps) {
        const size_t chunks = (size_t)chase_chunks(T);
        w.maps = reinterpret_cast<int32_t *>(p + w.bytes);
        w.bytes += align_up(sizeof(int32_t) * (size_t)B * chunks * S, 256);
        w.entries = reinterpret_cast<int32_t *>(p + w.bytes);
        w.bytes += align_up(sizeof(int32_t) * (size_t)B * chunks, 256);
    }
    w.arrive = nullptr;
    if (small::block_supported(S) && B <= 1024) {
        w.arrive = reinterpret_cast<int32_t *>(p + w.bytes);
        w.bytes += align_up(sizeof(int32_t) * (size_t)B * 8, 256);
    }
    return w;
}

// the time-resident path: history, tile map, row maxima, sorted lists / transposed matrix, cluster exchange
struct ResidentWorkspace {
    float *hist;
    float2 *sorted;
    float *tt;
    int32_t *row_range;
    int32_t *order;       // [B] this batch's items by descending length
    float *rowmax;        // [B][T] largest entry of every posterior row (left by the forward kernel for the backtrace)
    int32_t *lengths_hist;  // [T + 2] items per length (batches above resident::kMaxOrdered items only, else null)
    size_t lengths_hist_bytes;
    int32_t *tile_map;    // [kMaxGroupTiles] workgroup -> tile of the launch group (first batch's workspace)
    unsigned *stats;      // [128] scan statistics of the last launch group (first batch's workspace)
    // cluster form (first batch's workspace): exchange buffers of up to cus / 2 tiles
    float *xchg;          // [tiles][2][S4][16]
    unsigned *flags;      // [tiles][kMaxR] + 16 control words
    size_t flag_bytes;
    int SpP, NPOW;
    size_t bytes;
};

// The per-transition preparation of the time-resident routes (sorted + arranged rows, transposed matrix, row ranges)
// may live OUTSIDE the workspace (torbi_hip_viterbi_decode_batches_prepared): a caller that allocates a workspace per call
// -- the reference's own calling pattern, torbi/core.py:200-206 -- keeps 25 MB per matrix instead of rebuilding it
// (0.25 ms per call at 1440 states).  Set for the duration of one call on the calling thread.
thread_local void *g_preparation = nullptr;
thread_local size_t g_preparation_bytes = 0;
thread_local bool g_preparation_valid = false;      // holds this matrix's preparation (the caller's promise, or filled by this call)

// the serial number of the decode this host thread is launching: what its kernels raise their NaN / +inf alarms with
// (nonfinite.hpp; never 0, never repeated within a process: no alarm word has to be cleared between decodes)
static std::atomic<unsigned> g_serial_counter{(unsigned)(std::chrono::steady_clock::now().time_since_epoch().count() * 2654435761u)};
thread_local int t_serial = 0;          // of the decode this host thread is launching
inline int new_serial() {
    unsigned v;
    do { v = ++g_serial_counter; } while (v == 0u);
    t_serial = (int)v;
    return t_serial;
}
inline size_t preparation_bytes(int S) {
    const int Sp = (S + 15) / 16 * 16;
    return align_up(sizeof(float2) * (size_t)S * (Sp + pruned::kPad), 256) + align_up(sizeof(float) * (size_t)S * S, 256) +
           align_up(sizeof(int32_t) * 2 * (size_t)S, 256);
}

inline ResidentWorkspace carve_resident(void *base, int B, int T, int S, int cus) {
    ResidentWorkspace w;
    char *p = static_cast<char *>(base);
    const int Sp = (S + 15) / 16 * 16;
    w.SpP = Sp + pruned::kPad;
    w.NPOW = 64;
    while (w.NPOW < S) w.NPOW *= 2;
    const size_t hist_bytes = history_bytes(B, T, S);
    const size_t sorted_bytes = align_up(sizeof(float2) * (size_t)S * w.SpP, 256);
    const size_t tt_bytes = align_up(sizeof(float) * (size_t)S * S, 256);
    const size_t range_bytes = align_up(sizeof(int32_t) * 2 * (size_t)S, 256);
    const size_t hist_len = B > resident::kMaxOrdered ? align_up(sizeof(int32_t) * ((size_t)T + 2), 256) : 0;
    const size_t order_bytes = align_up(sizeof(int32_t) * (size_t)B, 256) + sizeof(int32_t) * (kMaxGroupTiles + 128) + hist_len;
    const size_t ctiles = (size_t)std::max(cus / 2, 1);                 // a cluster launch holds at most this many tiles
    const size_t xchg_bytes = align_up(ctiles * resident::kSlots * resident::cluster_slot_bytes(S), 256);
    w.flag_bytes = align_up(sizeof(unsigned) * (2 * ctiles * resident::kMaxR + 16 + ctiles), 256);    // flags, control, failed, where
    w.hist = reinterpret_cast<float *>(p);
    p += hist_bytes;
    w.tile_map = reinterpret_cast<int32_t *>(p);   // ahead of the preparation: offsets depend on B and T only
    w.stats = reinterpret_cast<unsigned *>(w.tile_map + kMaxGroupTiles);
    w.order = w.tile_map + kMaxGroupTiles + 128;
    w.lengths_hist_bytes = hist_len;
    w.lengths_hist = hist_len ? reinterpret_cast<int32_t *>(p + order_bytes - hist_len) : nullptr;
    p += order_bytes;
    const size_t rowmax_bytes = align_up(sizeof(float) * (size_t)B * T, 256);
    w.rowmax = reinterpret_cast<float *>(p);
    p += rowmax_bytes;
    w.sorted = reinterpret_cast<float2 *>(p);
    w.tt = reinterpret_cast<float *>(p + sorted_bytes);
    w.row_range = reinterpret_cast<int32_t *>(p + sorted_bytes + tt_bytes);
    if (base && g_preparation && g_preparation_bytes >= preparation_bytes(S)) {      // (the caller keeps it: see above)
        char *q = static_cast<char *>(g_preparation);
        w.sorted = reinterpret_cast<float2 *>(q);
        w.tt = reinterpret_cast<float *>(q + sorted_bytes);
        w.row_range = reinterpret_cast<int32_t *>(q + sorted_bytes + tt_bytes);
    }
    p += sorted_bytes + tt_bytes + range_bytes;
    w.xchg = reinterpret_cast<float *>(p);
    w.flags = reinterpret_cast<unsigned *>(p + xchg_bytes);
    w.bytes = hist_bytes + order_bytes + rowmax_bytes + sorted_bytes + tt_bytes + range_bytes + xchg_bytes + w.flag_bytes;
    return w;
}

// the band route (band_forward.hpp): the time-resident layout's history, tile map, statistics and item order, and BEHIND that
// layout the batch's exchange buffers and (first batch of a launch) the tickets and the give-up flags of the tiles
inline bool band_shape(int S) { return S % 4 == 0 && resident::supported(S) && S <= 64 * band::kMaxBlocks * band::kMaxR; }
struct BandWorkspace {
    ResidentWorkspace base;
    char *xchg;
    size_t xchg_bytes;
    unsigned *words;      // [16 ..] failed[kMaxGroupTiles], behind them [8] tickets per launch of the group
    float *tpack;         // whole tiles (band_tile_forward.hpp): the band as the lanes read it
    size_t bytes;
};
constexpr size_t kBandWords = 16 + 2 * (size_t)kMaxGroupTiles + 64;
inline BandWorkspace carve_band(void *base, int B, int T, int S, int cus) {
    BandWorkspace w;
    void *const kept = g_preparation;
    g_preparation = nullptr;                     // (the band route keeps nothing in a caller's preparation buffer)
    w.base = carve_resident(base, B, T, S, cus);
    g_preparation = kept;
    char *p = static_cast<char *>(base) + w.base.bytes;
    w.xchg_bytes = align_up(band::xchg_bytes(B, S), 256);
    w.xchg = p;
    w.words = reinterpret_cast<unsigned *>(p + w.xchg_bytes);
    const size_t word_bytes = align_up(sizeof(unsigned) * kBandWords, 256);
    w.tpack = reinterpret_cast<float *>(p + w.xchg_bytes + word_bytes);
    w.bytes = w.base.bytes + w.xchg_bytes + word_bytes + align_up(band::tile_pack_bytes_max(S), 256);
    return w;
}

// small batches (B <= 16): history + sorted rows (small_batch_forward.hpp)
struct RowsWorkspace {
    float *hist;
    float2 *sorted;
    int32_t *row_range;
    float *rowmax;     // [B][T] largest entry of every posterior row (step_rows_sorted_kernel leaves it for the backtrace)
    int SpP, NPOW;
    size_t bytes;
};

inline RowsWorkspace carve_rows(void *base, int B, int T, int S) {
    RowsWorkspace w;
    char *p = static_cast<char *>(base);
    const int Sp = (S + 15) / 16 * 16;
    w.SpP = Sp + pruned::kPad;
    w.NPOW = 64;
    while (w.NPOW < S) w.NPOW *= 2;
    const size_t sorted_bytes = align_up(sizeof(float2) * (size_t)S * w.SpP, 256);
    w.hist = reinterpret_cast<float *>(p);
    p += history_bytes(B, T, S);
    w.sorted = reinterpret_cast<float2 *>(p);
    w.row_range = reinterpret_cast<int32_t *>(p + sorted_bytes);
    const size_t range_bytes = align_up(sizeof(int32_t) * 2 * (size_t)S, 256);
    w.rowmax = reinterpret_cast<float *>(p + sorted_bytes + range_bytes);
    w.bytes = history_bytes(B, T, S) + sorted_bytes + range_bytes + align_up(sizeof(float) * (size_t)B * T, 256);
    return w;
}

struct DenseWorkspace {
    dense::Plan plan;
    float *hist;       // [B][T][S]      posterior history (replaces the int32 trellis)
    float *panel[2];   // [n_bt][Kp][BT] posterior panels (ping-pong)
    float *trp;        // [n_jt][Kp][W]  packed transition panels
    int32_t *chunks;   // [n_jt][NCH+1]  per-tile lists of chunks that are not all -inf
    int32_t *ranges;   // [S][2] finite range of every transition row + [64] (first word: the widest row window)
    size_t bytes;
};

// TORBI_HIP_BL=4|8 forces the batch-tile width of the dense path (experiments; default heuristic)
inline int bl_override() {
    static const int v = [] {
        const char *e = getenv("TORBI_HIP_BL");
        const int x = e ? atoi(e) : 0;
        return (x == 4 || x == 8) ? x : 0;
    }();
    return v;
}

// TORBI_HIP_NW=8|16 forces the waves per workgroup of the 8x6 dense tile (experiments)
inline int nw_override() {
    static const int v = [] {
        const char *e = getenv("TORBI_HIP_NW");
        const int x = e ? atoi(e) : 0;
        return (x == 8 || x == 16) ? x : 0;
    }();
    return v;
}

inline DenseWorkspace carve_dense(void *base, int B, int T, int S, int cus) {
    DenseWorkspace w;
    w.plan = dense::make_plan(B, S, cus, bl_override(), nw_override());
    char *p = static_cast<char *>(base);
    const size_t panel_bytes = align_up(sizeof(float) * (size_t)w.plan.n_bt * w.plan.Kp * w.plan.BT, 256);
    const size_t trp_bytes = align_up(sizeof(float) * (size_t)w.plan.n_jt * w.plan.Kp * w.plan.W, 256);
    const size_t list_bytes = align_up(sizeof(int32_t) * (size_t)w.plan.n_jt * (w.plan.NCH + 1), 256);
    w.hist = reinterpret_cast<float *>(p);
    p += history_bytes(B, T, S);
    w.panel[0] = reinterpret_cast<float *>(p);
    w.panel[1] = reinterpret_cast<float *>(p + panel_bytes);
    w.trp = reinterpret_cast<float *>(p + 2 * panel_bytes);
    w.chunks = reinterpret_cast<int32_t *>(p + 2 * panel_bytes + trp_bytes);
    w.ranges = reinterpret_cast<int32_t *>(p + 2 * panel_bytes + trp_bytes + list_bytes);
    w.bytes = history_bytes(B, T, S) + 2 * panel_bytes + trp_bytes + list_bytes +
              align_up(sizeof(int32_t) * (2 * (size_t)S + 64), 256);
    return w;
}

// scratch a (B,T,S) problem needs on a device with `cus` compute units, whichever path runs ...
inline size_t layout_bytes(int B, int T, int S, int cus) {
    size_t need = carve(nullptr, B, T, S).bytes;
    if (use_dense(B, S)) need = std::max(need, carve_dense(nullptr, B, T, S, cus).bytes);
    if (resident::supported(S)) need = std::max(need, carve_resident(nullptr, B, T, S, cus).bytes);
    if (rowscan::supported(B, S)) need = std::max(need, carve_rows(nullptr, B, T, S).bytes);
    if (band_shape(S)) need = std::max(need, carve_band(nullptr, B, T, S, cus).bytes);
    return align_up(need, 256);
}
// ... plus the ROUTE RECORD behind every layout: the forward path the last decode with this workspace actually took
// (a Route), written on the stream by that decode and read ON THE DEVICE by torbi_hip_read_posterior /
// torbi_hip_scan_stats -- a batch decoded inside a launch group takes the group's route, not the one its own shape
// and flags would give it.
// ... and behind the record the two posterior rows per item of nonfinite::repair_kernel (NaN / +inf inputs: nonfinite.hpp)
inline size_t nonfinite_rows_bytes(int B, int S) { return align_up(sizeof(float) * 2 * (size_t)B * S, 256); }
inline size_t need_bytes(int B, int T, int S, int cus) { return layout_bytes(B, T, S, cus) + 256 + nonfinite_rows_bytes(B, S); }
inline int32_t *route_record(const void *workspace, int B, int T, int S, int cus) {
    return reinterpret_cast<int32_t *>(static_cast<char *>(const_cast<void *>(workspace)) + layout_bytes(B, T, S, cus));
}
__global__ void stamp_route_kernel(int32_t *record, int route) { *record = route; }
__global__ void fill_pair_kernel(int32_t *pair, int a, int b) { pair[0] = a; pair[1] = b; }
inline hipError_t stamp_route(void *workspace, int B, int T, int S, int cus, Route route, hipStream_t s) {
    // (a one-thread kernel: hipMemsetD32Async costs ~0.17 ms per call on this stack)
    hipLaunchKernelGGL(stamp_route_kernel, dim3(1), dim3(1), 0, s, route_record(workspace, B, T, S, cus), (int)route);
    return hipGetLastError();
}

// ... on the device with the most demanding plan (devices of one node are normally identical)
inline size_t need_bytes_any_device(int B, int T, int S) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return need_bytes(B, T, S, 256);
    size_t need = 0;
    for (int d = 0; d < n && d < kMaxDevices; ++d) {
        const int cus = cu_count(d);
        bool seen = false;
        for (int e = 0; e < d; ++e) seen = seen || cu_count(e) == cus;
        if (!seen) need = std::max(need, need_bytes(B, T, S, cus));
    }
    return need;
}

int check_args(const void *a, const void *b, const void *c, const void *d, const void *e,
               const void *ws, size_t ws_bytes, int B, int T, int S, int device) {
    if (B < 0 || T < 1 || S < 1) return TORBI_HIP_EINVAL;
    if (B == 0) return TORBI_HIP_OK;
    if (!a || !b || !c || !d || !e || !ws) return TORBI_HIP_EINVAL;
    if ((size_t)B * T * S > (size_t)1 << 40) return TORBI_HIP_ERANGE;
    if (ws_bytes < need_bytes(B, T, S, cu_count(device))) return TORBI_HIP_EWORKSPACE;
    return TORBI_HIP_OK;
}

// ---- generic path ---------------------------------------------------------------------
hipError_t launch_forward(const float *obs, const int32_t *frames, const float *trans,
                          const float *init, const Workspace &w, int B, int T, int S,
                          hipStream_t stream, int *launches) {
    {
        const size_t n = (size_t)B * S;
        const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(init_posterior_kernel, dim3(grid), dim3(256), 0, stream, obs, init,
                           w.post[0], B, T, S);
    }
    // one workgroup per (rows, item).  (Rounds 1-3 had a 64 x 64 tile kernel here for many items over fewer than 64
    // states; those shapes are decoded in one launch by small_states.hpp now.)
    int n = 0;
    for (int t = 1; t < T; ++t) {
        const float *pc = w.post[(t - 1) & 1];
        float *pn = w.post[t & 1];
        // (the item index is gridDim.y, which holds at most 65535: a larger batch -- S == 1 under AUTO, DENSE named below
        // 64 states -- takes its timestep in slices of that many items)
        for (int b0 = 0; b0 < B; b0 += 65535) {
            const int nb = std::min(65535, B - b0);
            if (S % 4 == 0 && S >= 256 && (reinterpret_cast<uintptr_t>(trans) & 15) == 0) {
                dim3 grid((S + 3) / 4, nb);
                hipLaunchKernelGGL(step_rows4_kernel, grid, dim3(256), sizeof(float) * (size_t)S,
                                   stream, obs, frames, trans, pc, pn, w.trellis, B, T, S, t, b0);
            } else {
                dim3 grid((S + kRowsPerBlock - 1) / kRowsPerBlock, nb);
                hipLaunchKernelGGL(step_rows_kernel, grid, dim3(256), sizeof(float) * (size_t)S,
                                   stream, obs, frames, trans, pc, pn, w.trellis, B, T, S, t, b0);
            }
        }
        ++n;
    }
    if (launches) *launches = n;
    return hipGetLastError();
}

// The held-matrix kernel's instantiation for S states: kernel, grid, block.
struct HeldLaunch { const void *fn; int grid, block; };
inline HeldLaunch held_launch(int S) {
    const int K = (S + held::threads(S) - 1) / held::threads(S);
    const void *fn;
    if (S <= held::kSmallS)
        fn = K == 1 ? (const void *)&held::held_forward_kernel<1, 8, 512, true>
           : K == 2 ? (const void *)&held::held_forward_kernel<2, 8, 512, true>
           : K == 3 ? (const void *)&held::held_forward_kernel<3, 8, 512, true>
                    : (const void *)&held::held_forward_kernel<4, 8, 512, true>;
    else
        fn = K == 3 ? (const void *)&held::held_forward_kernel<3, 16, 1024, false>
                    : (const void *)&held::held_forward_kernel<4, 16, 1024, false>;
    return HeldLaunch{fn, held::workgroups(S), held::block_threads(S)};
}
// Can every workgroup of that launch be resident at once on `device`?  The runtime's occupancy answer for the very
// kernel (registers, LDS, waves), queried once per (kernel, device) -- held::supported()'s "two workgroups per compute
// unit up to 2048 states" is an assumption about this build on an MI355X, this is the check.
inline bool held_resident(int S, int device, int cus) {
    struct Known { const void *fn; int device; int per_cu; };
    static std::mutex mu;
    static std::vector<Known> known;
    const HeldLaunch h = held_launch(S);
    std::lock_guard<std::mutex> hold(mu);
    for (auto &k : known)
        if (k.fn == h.fn && k.device == device) return h.grid <= k.per_cu * cus;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, h.fn, h.block, 0) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 0;
    }
    known.push_back(Known{h.fn, device, per_cu});
    return h.grid <= per_cu * cus;
}

// the same outputs from ONE launch: the time loop inside the kernel, the matrix in registers (held_matrix_forward.hpp)
hipError_t launch_held_forward(const float *obs, const int32_t *frames, const float *trans, const float *init,
                               const Workspace &w, int B, int T, int S, hipStream_t stream, int *launches) {
    {
        const size_t n = (size_t)B * S;
        const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(held::prepare_kernel, dim3(grid), dim3(256), 0, stream, obs, init, w.post[0], w.xchg, w.control,
                           B, T, S);
    }
    if (launches) *launches = 1;
    if (T < 2) return hipGetLastError();
    // how long a workgroup waits for the others before it gives up (ticks of the 100 MHz wall clock): 20 x T x 2.5 us, at
    // least 2 ms -- 25 ms for 500 frames, against the ~1 ms the launch takes when it is resident.  TORBI_HIP_HELD_WAIT_US
    // overrides; TORBI_HIP_HELD_SPIN_LIMIT (polls of ~1 us, the round-3 knob; 0 forces the repair path in the tests) too.
    unsigned long long wait_ticks = std::max<unsigned long long>(200000ull, 5000ull * (unsigned long long)T);
    if (const char *us = getenv("TORBI_HIP_HELD_WAIT_US")) wait_ticks = 100ull * strtoull(us, nullptr, 10);
    else if (const char *polls = getenv("TORBI_HIP_HELD_SPIN_LIMIT")) wait_ticks = 100ull * strtoull(polls, nullptr, 10);
    const dim3 grid(held::workgroups(S)), block(held::block_threads(S));
    const int K = (S + held::threads(S) - 1) / held::threads(S);
#define TORBI_HELD(K_, R_, N_)                                                                                       \
    hipLaunchKernelGGL((held::held_forward_kernel<K_, R_, N_, (N_ < 1024)>), grid, block, 0, stream, obs, frames, trans, \
                       w.post[0], w.post[1], w.trellis, w.xchg, w.control, B, T, S, wait_ticks)
    if (S <= held::kSmallS) {
        if (K == 1) TORBI_HELD(1, 8, 512);
        else if (K == 2) TORBI_HELD(2, 8, 512);
        else if (K == 3) TORBI_HELD(3, 8, 512);
        else TORBI_HELD(4, 8, 512);
    } else if (K == 3) {
        TORBI_HELD(3, 16, 1024);
    } else {
        TORBI_HELD(4, 16, 1024);
    }
#undef TORBI_HELD
    // does nothing unless a workgroup above gave up waiting (held_matrix_forward.hpp)
    hipLaunchKernelGGL(held::repair_kernel, dim3(B), dim3(1024), 2 * sizeof(float) * (size_t)S, stream, obs, frames, trans, init,
                       w.post[0], w.post[1], w.trellis, w.control, B, T, S);
    return hipGetLastError();
}

hipError_t launch_finalize(const int32_t *frames, const Workspace &w, int32_t *out, int B, int T,
                           int S, hipStream_t stream) {
    if (w.maps) {        // a handful of long sequences: the chase in parallel (chase_*_kernel above)
        const int chunks = chase_chunks(T);
        hipLaunchKernelGGL(chase_maps_kernel, dim3(chunks, B, (S + 255) / 256), dim3(256), 0, stream, frames, w.trellis, w.maps,
                           B, T, S, chunks);
        hipLaunchKernelGGL(chase_entries_kernel, dim3(B), dim3(64), 0, stream, w.post[0], w.post[1], frames, w.trellis, w.maps,
                           w.entries, out, B, T, S, chunks);
        hipLaunchKernelGGL(chase_chunks_kernel, dim3(chunks, B), dim3(64), 0, stream, frames, w.trellis, w.entries, out,
                           B, T, S, chunks);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(64), 0, stream, w.post[0], w.post[1],
                       frames, w.trellis, out, B, T, S);
    return hipGetLastError();
}

// ---- up to 64 states: one wavefront per sequence, one launch per decode (small_states.hpp) ----------------------------
// The value-only form (no backpointers, lazy argmax on the path, the matrix shared through the LDS) from 32 padded states
// and 2 x compute-units sequences up -- tools/small_value_probe.py, ms per decode, backpointers / value-only:
// 512 x 500 x 40 0.259 / 0.233, 512 x 500 x 64 0.343 / 0.294, 4096 x 500 x 64 0.848 / 0.713, 4096 x 500 x 40 0.553 / 0.450;
// below 32 states the walk back costs more than the cells save (512 x 500 x 3 0.121 / 0.171, 8192 x 500 x 16 0.468 / 0.495).
// TORBI_HIP_SMALL_VALUE=0|1 forces either form (read per launch: the tests switch it).
inline bool small_value_form(int B, int S, int cus) {
    if (const char *e = getenv("TORBI_HIP_SMALL_VALUE")) return atoi(e) != 0;
    return small::padded_states(S) >= 32 && (long long)B >= 2ll * cus;
}
template <int SP, int CH>
hipError_t launch_small_as(const float *obs, const int32_t *frames, const float *trans, const float *init, const Workspace &w,
                           int32_t *out, int32_t *record, int B, int T, int S, hipStream_t stream, bool value_form) {
    if (value_form) {
        TORBI_NOTE_KERNEL("small::decode_value_kernel<%d, %d>", SP, CH);
        hipLaunchKernelGGL((small::decode_value_kernel<SP, CH>), dim3((B + 3) / 4), dim3(256), 0, stream, obs, frames, trans, init,
                           out, reinterpret_cast<float *>(w.trellis), w.post[0], w.post[1], record, (int)ROUTE_SMALL, B, T, S, t_serial);
        return hipGetLastError();
    }
    TORBI_NOTE_KERNEL("small::decode_kernel<%d, %d>", SP, CH);
    hipLaunchKernelGGL((small::decode_kernel<SP, CH>), dim3(B), dim3(64), 0, stream, obs, frames, trans, init, out,
                       reinterpret_cast<uint32_t *>(w.trellis), w.post[0], w.post[1], record, (int)ROUTE_SMALL, B, T, S, t_serial);
    return hipGetLastError();
}
hipError_t launch_small(const float *obs, const int32_t *frames, const float *trans, const float *init, const Workspace &w,
                        int32_t *out, int32_t *record, int B, int T, int S, hipStream_t stream, int *launches, int cus = 256) {
    if (launches) *launches += 1;
    const bool v = small_value_form(B, S, cus);
    switch (small::padded_states(S)) {
        case 4: return launch_small_as<4, 16>(obs, frames, trans, init, w, out, record, B, T, S, stream, v);
        case 8: return launch_small_as<8, 16>(obs, frames, trans, init, w, out, record, B, T, S, stream, v);
        case 16: return launch_small_as<16, 16>(obs, frames, trans, init, w, out, record, B, T, S, stream, v);
        case 24: return launch_small_as<24, 8>(obs, frames, trans, init, w, out, record, B, T, S, stream, v);
        case 32: return launch_small_as<32, 8>(obs, frames, trans, init, w, out, record, B, T, S, stream, v);
        case 40: return launch_small_as<40, 8>(obs, frames, trans, init, w, out, record, B, T, S, stream, v);
        case 48: return launch_small_as<48, 4>(obs, frames, trans, init, w, out, record, B, T, S, stream, v);
        case 56: return launch_small_as<56, 4>(obs, frames, trans, init, w, out, record, B, T, S, stream, v);
        default: return launch_small_as<64, 4>(obs, frames, trans, init, w, out, record, B, T, S, stream, v);
    }
}

// 65 .. 256 states: the value-only workgroup kernel + backtrace launches of their own (small_states.hpp, block_value_kernel)
inline int backtrace_segments(int items);
hipError_t launch_backtrace_on(const float *hist, const float *trans, const int32_t *frames, int32_t *out,
                               int B, int T, int S, hipStream_t stream, const int32_t *ranges, const int32_t *widest);
template <int PQ, int L>
hipError_t launch_block_value_as(const float *obs, const int32_t *frames, const float *trans, const float *init,
                                 const Workspace &w, int32_t *record, int B, int T, int S, hipStream_t stream, int cus) {
    const int NB = (S + 63) / 64;
    // two sequences per workgroup once the compute units are full without it: a 9- or 16-wave workgroup has a unit to
    // itself (512 x 500 x 256: 1.07 -> 1.03 ms, 2048 x 200 x 192: 1.33 -> 1.19), several 4-wave workgroups share one and
    // overlap anyway (512 x 500 x 128: 0.44 -> 0.64, but 4096 x 200 x 128: 0.95 -> 0.89).  TORBI_HIP_BLOCK_PAIRS=0 / 1 overrides
    const char *env = getenv("TORBI_HIP_BLOCK_PAIRS");
    const bool pairs = env ? atoi(env) != 0 : B > cus * (NB * PQ >= 9 ? 1 : 8);
    TORBI_NOTE_KERNEL("small::block_value_kernel<%d, %d, %d>", PQ, L, pairs ? 2 : 1);
    if (pairs)
        hipLaunchKernelGGL((small::block_value_kernel<PQ, L, 2>), dim3((B + 1) / 2), dim3(64 * NB * PQ), 0, stream, obs, frames,
                           trans, init, reinterpret_cast<float *>(w.trellis), w.post[0], w.post[1], record, (int)ROUTE_SMALL, B,
                           T, S, NB, t_serial);
    else
        hipLaunchKernelGGL((small::block_value_kernel<PQ, L, 1>), dim3(B), dim3(64 * NB * PQ), 0, stream, obs, frames, trans,
                           init, reinterpret_cast<float *>(w.trellis), w.post[0], w.post[1], record, (int)ROUTE_SMALL, B, T, S,
                           NB, t_serial);
    return hipGetLastError();
}
hipError_t launch_block(const float *obs, const int32_t *frames, const float *trans, const float *init, const Workspace &w,
                        int32_t *out, int32_t *record, int B, int T, int S, hipStream_t stream, int *launches, int cus) {
    const bool narrow = small::block_row_registers(S) == 48;
    if (launches) *launches += 2;
    hipError_t e;
    switch (small::block_splits(S)) {
        case 2: e = narrow ? launch_block_value_as<2, 48>(obs, frames, trans, init, w, record, B, T, S, stream, cus)
                           : launch_block_value_as<2, 64>(obs, frames, trans, init, w, record, B, T, S, stream, cus); break;
        case 3: e = narrow ? launch_block_value_as<3, 48>(obs, frames, trans, init, w, record, B, T, S, stream, cus)
                           : launch_block_value_as<3, 64>(obs, frames, trans, init, w, record, B, T, S, stream, cus); break;
        default: e = launch_block_value_as<4, 64>(obs, frames, trans, init, w, record, B, T, S, stream, cus);
    }
    if (e != hipSuccess) return e;
    const float *hist = reinterpret_cast<const float *>(w.trellis);
    const bool vec = (S % 4 == 0) && ((reinterpret_cast<uintptr_t>(trans) & 15) == 0);
    const int K = (vec && w.arrive) ? std::min(backtrace_segments(B), 8) : 1;
    if (K > 1) {      // few paths: speculative segments (lazy_backtrace.hpp, chase_segment)
        hipLaunchKernelGGL(lazy::segment_rows_kernel<1>, dim3(B * K), dim3(64), 0, stream, hist, trans, frames, out, B, T, S, K,
                           w.arrive);
        hipLaunchKernelGGL(lazy::stitch_rows_kernel<1>, dim3(B), dim3(64), 0, stream, hist, trans, frames, out, B, T, S, K, w.arrive);
        return hipGetLastError();
    }
    return launch_backtrace_on(hist, trans, frames, out, B, T, S, stream, nullptr, nullptr);
}

// ---- dense path -----------------------------------------------------------------------
template <int BL, int JL, int NW, int KC, int MSL>
hipError_t launch_dense_steps(const float *obs, const int32_t *frames, const DenseWorkspace &w,
                              int B, int T, int S, hipStream_t stream, int *launches, unsigned *clock_out) {
    const dense::Plan &pl = w.plan;
    const size_t lds = dense::lds_bytes<BL, JL, NW, KC, MSL>();
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&dense::step_dense_kernel<BL, JL, NW, KC, MSL>), lds);
    if (e != hipSuccess) return e;
    TORBI_NOTE_KERNEL("dense::step_dense_kernel<%d, %d, %d, %d, %d>", BL, JL, NW, KC, MSL);
    const int ntiles = pl.n_bt * pl.n_jt;
    const int grid = 8 * ((ntiles + 7) / 8);
    int n = 0;
    for (int t = 1; t < T; ++t) {
        hipLaunchKernelGGL((dense::step_dense_kernel<BL, JL, NW, KC, MSL>), dim3(grid), dim3(64 * NW), lds, stream, obs,
                           frames, w.trp, w.panel[(t - 1) & 1], w.panel[t & 1], w.hist, w.chunks, B, T, S,
                           t, pl.n_bt, pl.n_jt, pl.JT, pl.Kp, pl.NCH, pl.RB, clock_out);
        ++n;
    }
    if (launches) *launches = n;
    return hipGetLastError();
}

hipError_t launch_dense_forward(const float *obs, const int32_t *frames, const float *trans,
                                const float *init, const DenseWorkspace &w, int B, int T, int S,
                                hipStream_t stream, int *launches, bool reuse, unsigned *clock_out) {
    const dense::Plan &pl = w.plan;
    if (!reuse) {      // per-transition preparation: packed panels + per-tile lists of chunks that are not all -inf
        hipLaunchKernelGGL(dense::pack_transition_kernel, dim3((pl.Kp + 63) / 64, pl.n_jt), dim3(256), 0,
                           stream, trans, w.trp, S, pl.JT, pl.W, pl.Kp);
        hipLaunchKernelGGL(dense::build_chunk_lists_kernel, dim3(pl.n_jt), dim3(256),
                           sizeof(int) * (size_t)pl.NCH, stream, w.trp, w.chunks, S, pl.JT, pl.W, pl.Kp, pl.NCH,
                           pl.KC);
        // for the backtrace: the finite range of every row and the widest of them (lazy_backtrace.hpp)
        hipLaunchKernelGGL(stamp_route_kernel, dim3(1), dim3(1), 0, stream, w.ranges + 2 * (size_t)S, 0);
        hipLaunchKernelGGL(lazy::row_ranges_kernel, dim3(S), dim3(64), 0, stream, trans, w.ranges, w.ranges + 2 * (size_t)S, S);
    }
    {
        const size_t n = (size_t)pl.n_bt * pl.BT * pl.Kp;
        const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(dense::init_panels_kernel, dim3(grid), dim3(256), 0, stream, obs, init,
                           w.panel[0], w.panel[1], w.hist, B, T, S, pl.n_bt, pl.BT, pl.Kp);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
#define TORBI_DENSE_CASE(BL_, JL_, NW_, KC_, MSL_)                                             \
    if (pl.BL == BL_ && pl.JL == JL_ && pl.NW == NW_ && pl.KC == KC_ && pl.MSL == MSL_)         \
        return launch_dense_steps<BL_, JL_, NW_, KC_, MSL_>(obs, frames, w, B, T, S, stream, launches, clock_out)
    TORBI_DENSE_CASE(8, 6, 8, 12, 8);
    TORBI_DENSE_CASE(8, 6, 16, 6, 8);
    TORBI_DENSE_CASE(8, 4, 8, 12, 8);
    TORBI_DENSE_CASE(8, 2, 8, 12, 8);
    TORBI_DENSE_CASE(4, 6, 8, 12, 8);
    TORBI_DENSE_CASE(4, 4, 8, 12, 8);
    TORBI_DENSE_CASE(4, 2, 8, 12, 8);
#undef TORBI_DENSE_CASE
    return hipErrorInvalidValue;
}

// per-transition preparation shared by the pruned and the time-resident paths: descending rows with their
// prev-states as byte offsets into a [prev-state][items] posterior tile, block arrangement, transposed copy
void launch_list_preparation(const float *trans, float2 *sorted, int32_t *row_range, float *tt, int S, int SpP, int NPOW,
                             int items_per_tile, hipStream_t stream) {
    hipLaunchKernelGGL(pruned::sort_rows_kernel, dim3(S), dim3(256), sizeof(float) * 2 * (size_t)NPOW, stream, trans,
                       sorted, row_range, S, SpP, NPOW, items_per_tile * 4);
    if (items_per_tile == pruned::kNB) {
        const int n = (S / 4) * (SpP / pruned::kBlk);
        hipLaunchKernelGGL(pruned::arrange_blocks_kernel<4>, dim3((n + 63) / 64), dim3(64), 0, stream, sorted, S, SpP);
    } else {
        const int n = (S / 8) * (SpP / pruned::kBlk);
        hipLaunchKernelGGL(pruned::arrange_blocks_kernel<8>, dim3((n + 63) / 64), dim3(64), 0, stream, sorted, S, SpP);
    }
    hipLaunchKernelGGL(pruned::transpose_kernel, dim3((S + 31) / 32, (S + 31) / 32), dim3(256), 0, stream, trans, tt, S);
}

// (`ranges` / `widest`: the finite range of every transition row and the widest window, or null; with them a banded
// matrix is decoded by backtrace_ranged_kernel and the whole-row kernel returns at once -- decided on the device)
hipError_t launch_backtrace_on(const float *hist, const float *trans, const int32_t *frames, int32_t *out,
                               int B, int T, int S, hipStream_t stream, const int32_t *ranges = nullptr,
                               const int32_t *widest = nullptr) {
    const bool vec = (S % 4 == 0) && ((reinterpret_cast<uintptr_t>(trans) & 15) == 0);
    if (vec && S <= 256 * 16) {
        if (!ranges) widest = nullptr;
#define TORBI_BT_CASE(NQ_)                                                                          \
    if (S <= 256 * NQ_) {                                                                           \
        if (ranges)                                                                                 \
            hipLaunchKernelGGL(lazy::backtrace_ranged_kernel<NQ_>, dim3(B), dim3(64), 0, stream, hist, trans, ranges, \
                               widest, frames, out, B, T, S);                                       \
        hipLaunchKernelGGL(lazy::backtrace_prefetch_kernel<NQ_>, dim3(B), dim3(64), 0, stream, hist, \
                           trans, frames, out, B, T, S, widest);                                    \
        return hipGetLastError();                                                                   \
    }
        TORBI_BT_CASE(2)
        TORBI_BT_CASE(6)
        TORBI_BT_CASE(16)
#undef TORBI_BT_CASE
    }
    if (vec)
        hipLaunchKernelGGL(lazy::backtrace_kernel<4>, dim3(B), dim3(64), 0, stream, hist, trans,
                           frames, out, B, T, S);
    else
        hipLaunchKernelGGL(lazy::backtrace_kernel<1>, dim3(B), dim3(64), 0, stream, hist, trans,
                           frames, out, B, T, S);
    return hipGetLastError();
}

// backtrace over the SORTED transition rows the pruned / time-resident forward pass prepared (lazy_backtrace.hpp):
// a path step reads the posterior row + 0.5-1 KB of list instead of the posterior row + a whole transition row.
// TORBI_HIP_BACKTRACE=rows selects the transition-row form everywhere (experiments).
inline bool backtrace_sorted_enabled() {
    static const bool v = [] {
        const char *e = getenv("TORBI_HIP_BACKTRACE");
        return !(e && e[0] == 'r');
    }();
    return v;
}

// (`rowmax`: the row maxima the forward pass left, or null: with them the walk gathers instead of staging rows)
hipError_t launch_backtrace_sorted(const float *hist, const float2 *sorted, int SpP, int items_per_tile,
                                   const float *trans, const int32_t *frames, int32_t *out, int B, int T, int S,
                                   hipStream_t stream, const float *rowmax = nullptr) {
    if (!backtrace_sorted_enabled() || S % 4 != 0 || S > 256 * 16)
        return launch_backtrace_on(hist, trans, frames, out, B, T, S, stream);
    const int shift = items_per_tile == pruned::kNB ? 6 : 5;      // list offsets = prev-state * 4 * items per tile
    const size_t lds = sizeof(float) * (size_t)S;
    static const bool gather = [] {
        const char *e = getenv("TORBI_HIP_BACKTRACE_GATHER");
        return !e || atoi(e) != 0;
    }();
    if (rowmax && gather) {
#define TORBI_BTG_CASE(NQ_)                                                                                     \
        if (S <= 256 * NQ_) {                                                                                   \
            hipLaunchKernelGGL(lazy::backtrace_gather_kernel<NQ_>, dim3(B), dim3(64), 0, stream, hist, rowmax, sorted, \
                               SpP, shift, frames, out, B, T, S);                                               \
            return hipGetLastError();                                                                           \
        }
        TORBI_BTG_CASE(2)
        TORBI_BTG_CASE(6)
        TORBI_BTG_CASE(8)
        TORBI_BTG_CASE(16)
#undef TORBI_BTG_CASE
    }
#define TORBI_BTS_CASE(NQ_)                                                                              \
    if (S <= 256 * NQ_) {                                                                                \
        hipLaunchKernelGGL(lazy::backtrace_sorted_kernel<NQ_>, dim3(B), dim3(64), lds, stream, hist, sorted, SpP, \
                           shift, frames, out, B, T, S);                                                 \
        return hipGetLastError();                                                                        \
    }
    TORBI_BTS_CASE(2)
    TORBI_BTS_CASE(6)
    TORBI_BTS_CASE(8)
    TORBI_BTS_CASE(16)
#undef TORBI_BTS_CASE
    return hipErrorInvalidValue;
}

// ---- time-resident path: several batches, one forward launch, one backtrace launch -------------------------
struct HostBatch {
    const float *obs;
    const int32_t *frames;
    int32_t *out;
    void *workspace;
    int B, T;
};

// seeds per item of the time-resident kernel: 3, or 1 with TORBI_HIP_FEW_SEEDS; TORBI_HIP_RESIDENT_KR=0|1|3 overrides
// both (experiments)
inline int resident_seeds(bool few) {
    static const int forced = [] {
        const char *e = getenv("TORBI_HIP_RESIDENT_KR");
        const int x = e ? atoi(e) : -1;
        return (x == 0 || x == 1 || x == 3) ? x : -1;
    }();
    return forced >= 0 ? forced : (few ? 1 : 3);
}
// seeds of a launch from the call's flags: FEW -> one, MANY -> three, neither -> one in the cluster form, three with whole tiles
inline bool few_seeds(unsigned flags, bool clusters) {
    if (flags & TORBI_HIP_FEW_SEEDS) return true;
    if (flags & TORBI_HIP_MANY_SEEDS) return false;
    return clusters;
}

template <int KW, int MAXP, int KR, bool CLUSTER, int NI>
hipError_t launch_resident_variant(const resident::Group &grp, const resident::Cluster &clu, int workgroups,
                                   const ResidentWorkspace &w, const float *init, int S, hipStream_t stream) {
    const size_t lds = resident::lds_bytes(S, KR + 1);
    const void *fn = reinterpret_cast<const void *>(&resident::resident_forward_kernel<KW, MAXP, true, KR, CLUSTER, NI>);
    hipError_t e = ensure_dynamic_lds(fn, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((resident::resident_forward_kernel<KW, MAXP, true, KR, CLUSTER, NI>), dim3(workgroups), dim3(64 * KW), lds,
                       stream, grp, clu, w.tt, w.sorted, init, S, w.SpP);
    TORBI_NOTE_KERNEL("resident::resident_forward_kernel<%d, %d, true, %d, %s, %d, false>", KW, MAXP, KR, CLUSTER ? "true" : "false", NI);
    return hipGetLastError();
}

// (seeds: 3, or 1 with TORBI_HIP_FEW_SEEDS; the experiment value 0 of TORBI_HIP_RESIDENT_KR runs as 1)
template <int KW, int MAXP, bool CLUSTER, int NI = 16>
hipError_t launch_resident_kernel(const resident::Group &grp, const resident::Cluster &clu, int workgroups,
                                  const ResidentWorkspace &w, const float *init, int S, hipStream_t stream, bool few) {
    return resident_seeds(few) == 3 ? launch_resident_variant<KW, MAXP, 3, CLUSTER, NI>(grp, clu, workgroups, w, init, S, stream)
                                    : launch_resident_variant<KW, MAXP, 1, CLUSTER, NI>(grp, clu, workgroups, w, init, S, stream);
}

// the repair launch behind a cluster launch: whole tiles, only where Group::only is set; ONE instance per seed count and
// tile size (eleven passes cover every supported state count)
template <int KR, int NI>
hipError_t launch_repair_variant(const resident::Group &grp, const resident::Cluster &clu, int tiles, const ResidentWorkspace &w,
                                 const float *init, int S, hipStream_t stream) {
    const size_t lds = resident::lds_bytes(S, KR + 1);
    const void *fn = reinterpret_cast<const void *>(&resident::resident_forward_kernel<12, 11, true, KR, false, NI, true>);
    hipError_t e = ensure_dynamic_lds(fn, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((resident::resident_forward_kernel<12, 11, true, KR, false, NI, true>), dim3(tiles), dim3(64 * 12), lds,
                       stream, grp, clu, w.tt, w.sorted, init, S, w.SpP);
    return hipGetLastError();
}
inline hipError_t launch_repair(const resident::Group &grp, const resident::Cluster &clu, int tiles, const ResidentWorkspace &w,
                                const float *init, int S, hipStream_t s, bool few) {
    const bool small = resident::tile_items(S) != resident::kNI;
    if (resident_seeds(few) == 3)
        return small ? launch_repair_variant<3, 8>(grp, clu, tiles, w, init, S, s) : launch_repair_variant<3, 16>(grp, clu, tiles, w, init, S, s);
    return small ? launch_repair_variant<1, 8>(grp, clu, tiles, w, init, S, s) : launch_repair_variant<1, 16>(grp, clu, tiles, w, init, S, s);
}

inline bool cluster_eight_waves() {             // TORBI_HIP_CLUSTER_KW8=0: the twelve-wave instances everywhere (experiments)
    const char *e = getenv("TORBI_HIP_CLUSTER_KW8");
    return !e || atoi(e) != 0;
}

// every workgroup owns a whole tile (resident_forward_kernel without clusters)
inline hipError_t launch_whole_tiles(const resident::Group &grp, const resident::Cluster &clu, int tiles,
                                     const ResidentWorkspace &w, const float *init, int S, hipStream_t s, bool few) {
    const int nrg = (S + resident::pass_rows(S) - 1) / resident::pass_rows(S);
    if (resident::tile_items(S) != resident::kNI) {       // 8-item tiles (2048 < S <= 4096)
        // (eight waves x 16 passes with 256 registers each measured slower than twelve x 8-11 at 168: 197 against 183 us per
        // timestep at 4096 states with a third of the units busy, equal on a full chip)
        if (nrg <= 96) return launch_resident_kernel<12, 8, false, 8>(grp, clu, tiles, w, init, S, s, few);
        return launch_resident_kernel<12, 11, false, 8>(grp, clu, tiles, w, init, S, s, few);
    }
    if (nrg <= 72) return launch_resident_kernel<12, 6, false>(grp, clu, tiles, w, init, S, s, few);
    if (nrg <= 96) return launch_resident_kernel<12, 8, false>(grp, clu, tiles, w, init, S, s, few);
    return launch_resident_kernel<12, 11, false>(grp, clu, tiles, w, init, S, s, few);
}

// segments per path of the backtrace behind a time-resident forward launch: 8 while the launch holds few paths (a wave per
// path leaves the chip idle and pays its steps' latency one after the other); 1 = whole paths, which a launch group's
// thousands of paths are walked as (bound by the bytes they move).  TORBI_HIP_BACKTRACE_SEGMENTS overrides (1 = off).
inline int backtrace_segments(int items) {
    const char *e = getenv("TORBI_HIP_BACKTRACE_SEGMENTS");         // (read per launch: the tests switch it)
    const int wanted = e ? atoi(e) : 8;
    if (wanted <= 1 || items > 1024) return 1;
    return std::max(2, std::min(wanted, 2048 / std::max(items, 1)));        // (items x K <= 2048 waves: 512 items -> 4)
}

// ---- NaN / +inf inputs (nonfinite.hpp): alarms raised with the decode's serial number, a repair launch behind the decode ----
// Ahead of a decode's forward launches: the matrix and the initial vector; the observations too for the routes whose
// forward kernels do not look at what they produce (a launch per timestep: generic, held, rows, dense).
inline hipError_t nonfinite_begin(const HostBatch *hb, int n, const float *trans, const float *init, int S, int cus, hipStream_t s,
                                  bool scan_observations, int reach_left = -1, int reach_right = -1, bool scan_matrix = true,
                                  float background = -INFINITY) {
    const int serial = new_serial();
    if (!scan_matrix) return hipSuccess;            // (the small-state kernels hold the whole matrix: they look themselves)
    nonfinite::Records recs{};
    recs.n = n;
    for (int k = 0; k < n; ++k) recs.record[k] = route_record(hb[k].workspace, hb[k].B, hb[k].T, S, cus);
    const size_t cells = (size_t)S * S;
    hipLaunchKernelGGL(nonfinite::matrix_kernel, dim3((unsigned)std::max<size_t>(1, std::min<size_t>(1024, (cells + 2047) / 2048))),
                       dim3(256), 0, s, trans, init, S, recs, serial, reach_left, reach_right, background);
    if (scan_observations)
        for (int k = 0; k < n; ++k) {
            const size_t count = (size_t)hb[k].B * hb[k].T * S;
            hipLaunchKernelGGL(nonfinite::observation_kernel,
                               dim3((unsigned)std::max<size_t>(1, std::min<size_t>(2048, (count + 4095) / 4096))), dim3(256), 0, s,
                               hb[k].obs, count, recs.record[k], serial);
        }
    return hipGetLastError();
}
// Behind the decode's backtrace: does nothing unless an alarm carries this decode's serial number.
inline hipError_t nonfinite_end(const HostBatch *hb, int n, const float *trans, const float *init, int S, int cus, hipStream_t s) {
    nonfinite::RepairJobs jobs{};
    jobs.n = n;
    int items = 0;
    for (int k = 0; k < n; ++k) {
        jobs.obs[k] = hb[k].obs;
        jobs.frames[k] = hb[k].frames;
        jobs.out[k] = hb[k].out;
        jobs.trellis[k] = static_cast<int32_t *>(hb[k].workspace);
        jobs.record[k] = route_record(hb[k].workspace, hb[k].B, hb[k].T, S, cus);
        jobs.rows[k] = reinterpret_cast<float *>(reinterpret_cast<char *>(jobs.record[k]) + 256);
        jobs.B[k] = hb[k].B;
        jobs.T[k] = hb[k].T;
        jobs.item0[k] = items;
        items += hb[k].B;
    }
    hipLaunchKernelGGL(nonfinite::repair_kernel, dim3(items), dim3(256), 0, s, jobs, trans, init, S, t_serial);
    return hipGetLastError();
}

// batches with B > 0 only; the preparation lives in the first batch's workspace
hipError_t run_resident(const HostBatch *hb, int n, const float *trans, const float *init, int S, int cus, hipStream_t s,
                        hipEvent_t *ev, int *launches, bool reuse, bool ascending = false, bool clusters = false,
                        bool few = false) {
    resident::Group grp{};
    resident::OrderJobs jobs{};
    jobs.ascending = ascending ? 1 : 0;
    grp.n = n;
    {
        const hipError_t ne = nonfinite_begin(hb, n, trans, init, S, cus, s, false);
        if (ne != hipSuccess) return ne;
    }
    grp.serial = t_serial;
    int tiles = 0, items = 0, widest = 0;
    for (int k = 0; k < n; ++k) {
        resident::Batch &b = grp.batch[k];
        const ResidentWorkspace wk = carve_resident(hb[k].workspace, hb[k].B, hb[k].T, S, cus);
        b.alarm = route_record(hb[k].workspace, hb[k].B, hb[k].T, S, cus) + nonfinite::kAlarmWord;
        b.obs = hb[k].obs;
        b.frames = hb[k].frames;
        b.out = hb[k].out;
        b.hist = wk.hist;
        b.order = wk.order;
        b.rowmax = wk.rowmax;
        jobs.job[k] = resident::OrderJob{hb[k].frames, wk.order, hb[k].B, hb[k].T, tiles, wk.lengths_hist, nullptr, 0};
        widest = std::max(widest, hb[k].B);
        b.B = hb[k].B;
        b.T = hb[k].T;
        b.tile0 = tiles;
        b.item0 = items;
        tiles += tiles_of(hb[k].B, S);
        items += hb[k].B;
    }
    const ResidentWorkspace w = carve_resident(hb[0].workspace, hb[0].B, hb[0].T, S, cus);
    if (tiles > kMaxGroupTiles) return hipErrorInvalidValue;
    grp.tile_map = w.tile_map;
    grp.stats = w.stats;
    jobs.stats = w.stats;
    jobs.n = n;
    jobs.tiles = tiles;
    jobs.tile_map = w.tile_map;
    // cluster form: R workgroups per tile (exchange buffers in the first batch's workspace, flags and tickets zeroed)
    const int R = clusters ? cluster_members(tiles, S, cus) : 1;
    unsigned *const control = w.flags + (size_t)std::max(cus / 2, 1) * resident::kMaxR;
    const char *wait_env = getenv("TORBI_HIP_CLUSTER_WAIT_US");         // (read per launch: the tests switch it)
    const unsigned long long wait_ticks = wait_env ? 100ull * strtoull(wait_env, nullptr, 10) : resident::kClusterWaitTicks;
    // (flags [ctiles][kMaxR], control [16], failed [ctiles], where [ctiles][kMaxR]: all zeroed by order_tiles_kernel)
    // Above 2048 states (8-item tiles) the sorted lists are what the launch fetches -- 134 MB at 4096 states, far beyond an
    // XCD's 4 MB L2, walked by every tile: 232 GB per 128 x 2000 x 4096 decode against 8.4 GB algorithmic
    // (profiles/r06_c5_pmc.json).  Member m of EVERY tile on one XCD keeps that member's rows' lists in its L2; the exchange
    // then crosses XCDs (write-through), which costs less than it saves there: 19.8 -> 17.3 us per timestep at 128 items,
    // 34.3 -> 30.6 at 256; up to 2048 states it loses 1-3 % (profiles/r06_c5_spread.txt).  TORBI_HIP_CLUSTER_SPREAD=0|1 overrides.
    const char *spread_env = getenv("TORBI_HIP_CLUSTER_SPREAD");
    const int spread = (R % 8 == 0 && (spread_env ? atoi(spread_env) != 0 : resident::tile_items(S) == 8)) ? 1 : 0;
    resident::Cluster clu{w.xchg, control + 16 + std::max(cus / 2, 1), tiles, control, control + 16, R, wait_ticks, spread};
    if (ev) (void)hipEventRecord(ev[0], s);
    for (int k = 0; k < n; ++k) {            // (order_items_kernel stamps the batches' route records)
        jobs.job[k].route_record = route_record(hb[k].workspace, hb[k].B, hb[k].T, S, cus);
        jobs.job[k].route = (int)(R > 1 ? ROUTE_CLUSTER : ROUTE_RESIDENT);
    }
    if (g_preparation) reuse = g_preparation_valid;       // the caller's buffer: the promise is about IT, whichever batch
    if (!reuse) launch_list_preparation(trans, w.sorted, w.row_range, w.tt, S, w.SpP, w.NPOW, resident::tile_items(S), s);
    if (g_preparation) g_preparation_valid = true;
    hipLaunchKernelGGL(resident::order_items_kernel, dim3((widest + 255) / 256, n), dim3(256), 0, s, jobs);
    for (int k = 0; k < n; ++k) {            // batches too large for the all-pairs ranking: counting sort over the lengths
        const resident::OrderJob &jb = jobs.job[k];
        if (jb.B <= resident::kMaxOrdered) continue;
        const ResidentWorkspace wk = carve_resident(hb[k].workspace, hb[k].B, hb[k].T, S, cus);
        hipError_t me = hipMemsetAsync(wk.lengths_hist, 0, wk.lengths_hist_bytes, s);
        if (me != hipSuccess) return me;
        hipLaunchKernelGGL(resident::order_large_count_kernel, dim3((jb.B + 255) / 256), dim3(256), 0, s, jb);
        hipLaunchKernelGGL(resident::order_large_scan_kernel, dim3(1), dim3(1024), 0, s, jb, jobs.ascending);
        hipLaunchKernelGGL(resident::order_large_place_kernel, dim3((jb.B + 255) / 256), dim3(256), 0, s, jb);
    }
    jobs.ni = resident::tile_items(S);
    jobs.flags = w.flags;
    jobs.nflags = R > 1 ? (int)(w.flag_bytes / sizeof(unsigned)) : 0;
    hipLaunchKernelGGL(resident::order_tiles_kernel, dim3((tiles + 255) / 256), dim3(256), 0, s, jobs);
    hipError_t e;
    if (ev) (void)hipEventRecord(ev[3], s);
    const int nrg = (S + resident::pass_rows(S) - 1) / resident::pass_rows(S);
    const bool small = resident::tile_items(S) != resident::kNI;       // 8-item tiles (2048 < S <= 4096)
    if (R > 1) {
        {       // the slots this launch uses start out absent (resident_forward.hpp, cluster_slot_bytes)
            const size_t granules = (size_t)tiles * resident::kSlots * resident::cluster_slot_bytes(S) / 16;
            hipLaunchKernelGGL(resident::absent_kernel, dim3((unsigned)std::min<size_t>((granules + 255) / 256, 2048)), dim3(256),
                               0, s, reinterpret_cast<uint4 *>(w.xchg), granules);
        }
        const int passes = ((nrg + R - 1) / R + 11) / 12;       // row groups of the largest share over 12 waves
        // (eight dispatch classes of R x ceil(tiles / 8) workgroups each: resident_forward.hpp, struct Cluster)
        const int grid = 8 * ((tiles + 7) / 8) * R;
        const int share = (nrg + R - 1) / R;                    // row groups of the largest share
        if (small && share <= 16 && cluster_eight_waves()) {
            // 8-item tiles, at most 16 row groups a member: EIGHT waves (256 registers each: the twelve-wave instances of
            // the 8-item tile spill 20-80 registers at 168, and every scratch reload waits for the write-through stores
            // ahead of it); 128 x 4096 (16 tiles x 16 members, 8 row groups each) kept four of twelve waves idle anyway
            if (share <= 8) e = launch_resident_kernel<8, 1, true, 8>(grp, clu, grid, w, init, S, s, few);
            else e = launch_resident_kernel<8, 2, true, 8>(grp, clu, grid, w, init, S, s, few);
        } else if (small) {
            if (passes <= 1) e = launch_resident_kernel<12, 1, true, 8>(grp, clu, grid, w, init, S, s, few);
            else if (passes <= 2) e = launch_resident_kernel<12, 2, true, 8>(grp, clu, grid, w, init, S, s, few);
            else if (passes <= 4) e = launch_resident_kernel<12, 4, true, 8>(grp, clu, grid, w, init, S, s, few);
            else e = launch_resident_kernel<12, 6, true, 8>(grp, clu, grid, w, init, S, s, few);
        } else if (passes <= 1) e = launch_resident_kernel<12, 1, true>(grp, clu, grid, w, init, S, s, few);
        else if (passes <= 2) e = launch_resident_kernel<12, 2, true>(grp, clu, grid, w, init, S, s, few);
        else if (passes <= 4) e = launch_resident_kernel<12, 4, true>(grp, clu, grid, w, init, S, s, few);
        else e = launch_resident_kernel<12, 6, true>(grp, clu, grid, w, init, S, s, few);
        // a cluster that could not complete in time (resident_forward.hpp: CLUSTER_WAIT_TICKS) has flagged its tile: the
        // launch behind decodes those tiles again, whole -- it returns at once wherever nothing was flagged (every run so far)
        if (e == hipSuccess) {
            resident::Group again = grp;
            again.only = clu.failed;
            e = launch_repair(again, clu, tiles, w, init, S, s, few);
        }
    } else {
        e = launch_whole_tiles(grp, clu, tiles, w, init, S, s, few);
    }
    if (launches) *launches = 1;
    if (ev) (void)hipEventRecord(ev[1], s);
    if (e != hipSuccess) return e;
    const bool vec = (S % 4 == 0) && ((reinterpret_cast<uintptr_t>(trans) & 15) == 0);
    const size_t row_lds = sizeof(float) * (size_t)S;
    // the backtrace gathers the posteriors the lists point at instead of staging whole rows in the LDS: half the bytes, and
    // faster from one batch (0.58 against 0.80 ms) to eight (1.86 against 3.11 ms; tools/backtrace_probe.py,
    // profiles/r03_backtrace_gather.txt).  TORBI_HIP_BACKTRACE_GATHER=0 brings the staging form back.
    static const bool gather = [] {
        const char *e = getenv("TORBI_HIP_BACKTRACE_GATHER");
        return !e || atoi(e) != 0;
    }();
    // few paths (one batch): every path in speculative segments, K waves per item, then one wave per item at the joints
    // (lazy_backtrace.hpp, chase_segment); the forward launch is done with the tile map, which holds the segments' ends
    const int K = backtrace_segments(items);
    if (S % 4 == 0 && backtrace_sorted_enabled() && gather && K > 1) {
        int32_t *const arrive = w.tile_map;
#define TORBI_SEGMENTED(NQ_)                                                                                                      \
        hipLaunchKernelGGL(resident::group_segment_gather_kernel<NQ_>, dim3(items * K), dim3(64), 0, s, grp, w.sorted, w.SpP, S, K, \
                           arrive);                                                                                               \
        hipLaunchKernelGGL(resident::group_stitch_gather_kernel<NQ_>, dim3(items), dim3(64), 0, s, grp, w.sorted, w.SpP, S, K, arrive)
        if (S <= 512) { TORBI_SEGMENTED(2); }
        else if (S <= 1536) { TORBI_SEGMENTED(6); }
        else if (S <= 2048) { TORBI_SEGMENTED(8); }
        else { TORBI_SEGMENTED(16); }
#undef TORBI_SEGMENTED
    } else if (S % 4 == 0 && backtrace_sorted_enabled() && gather) {
        if (S <= 512)
            hipLaunchKernelGGL(resident::group_backtrace_gather_kernel<2>, dim3(items), dim3(64), 0, s, grp, w.sorted, w.SpP, S);
        else if (S <= 1536)
            hipLaunchKernelGGL(resident::group_backtrace_gather_kernel<6>, dim3(items), dim3(64), 0, s, grp, w.sorted, w.SpP, S);
        else if (S <= 2048)
            hipLaunchKernelGGL(resident::group_backtrace_gather_kernel<8>, dim3(items), dim3(64), 0, s, grp, w.sorted, w.SpP, S);
        else
            hipLaunchKernelGGL(resident::group_backtrace_gather_kernel<16>, dim3(items), dim3(64), 0, s, grp, w.sorted, w.SpP, S);
    } else if (S % 4 == 0 && backtrace_sorted_enabled()) {
        if (S <= 512)
            hipLaunchKernelGGL(resident::group_backtrace_sorted_kernel<2>, dim3(items), dim3(64), row_lds, s, grp,
                               w.sorted, w.SpP, S);
        else if (S <= 1536)
            hipLaunchKernelGGL(resident::group_backtrace_sorted_kernel<6>, dim3(items), dim3(64), row_lds, s, grp,
                               w.sorted, w.SpP, S);
        else if (S <= 2048)
            hipLaunchKernelGGL(resident::group_backtrace_sorted_kernel<8>, dim3(items), dim3(64), row_lds, s, grp,
                               w.sorted, w.SpP, S);
        else
            hipLaunchKernelGGL(resident::group_backtrace_sorted_kernel<16>, dim3(items), dim3(64), row_lds, s, grp,
                               w.sorted, w.SpP, S);
    } else if (vec && S <= 512)
        hipLaunchKernelGGL(resident::group_backtrace_prefetch_kernel<2>, dim3(items), dim3(64), 0, s, grp, trans, S);
    else if (vec && S <= 1536)
        hipLaunchKernelGGL(resident::group_backtrace_prefetch_kernel<6>, dim3(items), dim3(64), 0, s, grp, trans, S);
    else if (vec && S <= 2048)
        hipLaunchKernelGGL(resident::group_backtrace_prefetch_kernel<8>, dim3(items), dim3(64), 0, s, grp, trans, S);
    else if (vec)
        hipLaunchKernelGGL(resident::group_backtrace_prefetch_kernel<16>, dim3(items), dim3(64), 0, s, grp, trans, S);
    else
        hipLaunchKernelGGL(resident::group_backtrace_kernel<1>, dim3(items), dim3(64), 0, s, grp, trans, S);
    {
        const hipError_t ne = nonfinite_end(hb, n, trans, init, S, cus, s);
        if (ne != hipSuccess) return ne;
    }
    if (ev) (void)hipEventRecord(ev[2], s);
    return hipGetLastError();
}


// ---- band route: several batches, one forward launch (band_forward.hpp), one backtrace launch ------------------------
inline int band_tiles(int B) { return (B + band::kNI - 1) / band::kNI; }

// batches with B > 0 only; tile map, statistics, tickets and give-up flags live in the first batch's workspace
// What the band route does with a launch group: whole tiles (band_tile_forward.hpp: one workgroup per tile, once the group
// has a tile for at least every other compute unit) or tiles split over R members (band_forward.hpp), `cap` tiles per
// launch -- every member of a launch resident at once, because they wait for each other inside it.
struct BandChoice {
    band::Plan pl;          // split form (pl.R members per tile); whole tiles: S, hl, hr and R = 1 only
    band::TilePlan tile;
    bool whole;
    int cap;                // tiles per launch (split form)
    float background;      // every entry outside the band: -inf, or ONE constant (whole tiles only: band_tile_forward.hpp)
};
inline bool choose_band(int S, int hl, int hr, int tiles, int cus, BandChoice &c, float background = -INFINITY) {
    c = BandChoice{};
    c.background = background;
    if (tiles < 1 || background != background || background == INFINITY) return false;
    const char *force = getenv("TORBI_HIP_BAND_FORM");            // experiments: "tile" / "split"
    const char *tw = getenv("TORBI_HIP_TILE_WAVES");              // experiments: 8 / 12 waves per workgroup
    const bool tile_ok = band::make_tile_plan(S, hl, hr, c.tile, tw ? atoi(tw) : 0) && !(force && force[0] == 's');
    // Whole tiles need a tile for (almost) every compute unit to pay: a whole-tile timestep takes ~5 x a split one (36 against
    // 7.4 us at 1440 states, reach 87) whatever the tile count, a split launch decodes 32 tiles at a time.  Measured, -inf
    // band, launch groups of 512-item batches (tools/pitch_tiny_probe.py): 4 batches = 128 tiles 56 M whole against 67 M split,
    // 8 batches 109 M against 72 M; the rates cross near 157 tiles.  Where there is no split form (a constant outside the band;
    // reaches or state counts it does not cover) the alternative is the cluster form (34-40 M): whole tiles from half the units up.
    band::Plan split{};
    const bool split_ok = !(force && force[0] == 't' && tile_ok) && band::make_plan(S, hl, hr, tiles, cus, split, background) &&
                          band::tiles_per_launch(split, cus) >= 1;
    // (a split launch decodes `per` tiles in ~1.6 / R of a whole-tile timestep: whole tiles once the split launches of the group
    // add up to more -- 8 * launches >= 5 * R, which is 8 * tiles >= 5 * CUs for groups that fill their launches)
    const int per = split_ok ? std::max(1, std::min(band::tiles_per_launch(split, cus), kMaxGroupTiles)) : 1;
    const bool enough = split_ok ? 8 * ((tiles + per - 1) / per) >= 5 * split.R : 2 * tiles >= cus;
    if (tile_ok && (enough || (force && force[0] == 't'))) {
        c.whole = true;
        c.pl.S = S; c.pl.hl = hl; c.pl.hr = hr; c.pl.R = 1;
        c.cap = tiles;
        c.tile.background = background;
        return true;
    }
    if (!split_ok) return false;
    c.pl = split;
    c.cap = std::min(band::tiles_per_launch(c.pl, cus), kMaxGroupTiles);
    return true;
}

hipError_t run_band(const HostBatch *hb, int n, const float *trans, const float *init, int S, const BandChoice &choice, int cus,
                    hipStream_t s, hipEvent_t *ev, int *launches, bool ascending) {
    const band::Plan &pl = choice.pl;
    resident::Group grp{};
    resident::OrderJobs jobs{};
    band::Exchange ex{};
    band::ClearJobs clear{};
    jobs.ascending = ascending ? 1 : 0;
    grp.n = n;
    {
        const hipError_t ne = nonfinite_begin(hb, n, trans, init, S, cus, s, false, pl.hl, pl.hr, true, choice.background);     // (and the band's promise)
        if (ne != hipSuccess) return ne;
    }
    grp.serial = t_serial;
    int tiles = 0, items = 0, widest = 0;
    size_t most = 0;
    for (int k = 0; k < n; ++k) {
        resident::Batch &b = grp.batch[k];
        const BandWorkspace wk = carve_band(hb[k].workspace, hb[k].B, hb[k].T, S, cus);
        b.alarm = route_record(hb[k].workspace, hb[k].B, hb[k].T, S, cus) + nonfinite::kAlarmWord;
        b.obs = hb[k].obs;
        b.frames = hb[k].frames;
        b.out = hb[k].out;
        b.hist = wk.base.hist;
        b.order = wk.base.order;
        b.rowmax = wk.base.rowmax;
        jobs.job[k] = resident::OrderJob{hb[k].frames, wk.base.order, hb[k].B, hb[k].T, tiles, wk.base.lengths_hist,
                                         route_record(hb[k].workspace, hb[k].B, hb[k].T, S, cus), (int)ROUTE_BAND};
        ex.xchg[k] = wk.xchg;
        clear.xchg[k] = wk.xchg;
        clear.bytes[k] = pl.R > 1 ? band::xchg_bytes(hb[k].B, S) : 0;
        most = std::max(most, clear.bytes[k]);
        widest = std::max(widest, hb[k].B);
        b.B = hb[k].B;
        b.T = hb[k].T;
        b.tile0 = tiles;
        b.item0 = items;
        tiles += band_tiles(hb[k].B);
        items += hb[k].B;
    }
    if (tiles > kMaxGroupTiles) return hipErrorInvalidValue;
    const BandWorkspace w = carve_band(hb[0].workspace, hb[0].B, hb[0].T, S, cus);
    grp.tile_map = w.base.tile_map;
    grp.stats = w.base.stats;
    jobs.stats = w.base.stats;
    jobs.n = n;
    jobs.tiles = tiles;
    jobs.tile_map = w.base.tile_map;
    jobs.ni = band::kNI;
    const int nlaunch = choice.whole ? 1 : (tiles + choice.cap - 1) / choice.cap;
    ex.failed = w.words + 16;
    unsigned *const tickets = w.words + 16 + kMaxGroupTiles;        // [8] per launch
    const char *wait_env = getenv("TORBI_HIP_CLUSTER_WAIT_US");         // (read per launch: the tests switch it)
    ex.wait_ticks = wait_env ? 100ull * strtoull(wait_env, nullptr, 10) : resident::kClusterWaitTicks;
    clear.words = w.words;
    clear.nwords = 16 + kMaxGroupTiles + 8 * nlaunch;
    clear.n = n;
    if (choice.whole)
        for (int k = 0; k < n; ++k) clear.bytes[k] = 0;
    if (ev) (void)hipEventRecord(ev[0], s);
    hipLaunchKernelGGL(resident::order_items_kernel, dim3((widest + 255) / 256, n), dim3(256), 0, s, jobs);
    for (int k = 0; k < n; ++k) {            // batches too large for the all-pairs ranking: counting sort over the lengths
        const resident::OrderJob &jb = jobs.job[k];
        if (jb.B <= resident::kMaxOrdered) continue;
        const BandWorkspace wk = carve_band(hb[k].workspace, hb[k].B, hb[k].T, S, cus);
        hipError_t me = hipMemsetAsync(wk.base.lengths_hist, 0, wk.base.lengths_hist_bytes, s);
        if (me != hipSuccess) return me;
        hipLaunchKernelGGL(resident::order_large_count_kernel, dim3((jb.B + 255) / 256), dim3(256), 0, s, jb);
        hipLaunchKernelGGL(resident::order_large_scan_kernel, dim3(1), dim3(1024), 0, s, jb, jobs.ascending);
        hipLaunchKernelGGL(resident::order_large_place_kernel, dim3((jb.B + 255) / 256), dim3(256), 0, s, jb);
    }
    hipLaunchKernelGGL(resident::order_tiles_kernel, dim3((tiles + 255) / 256), dim3(256), 0, s, jobs);
    hipError_t e;
    if (choice.whole) {
        // whole tiles: the band packed in the lanes' reading order (1 MB at 1440 states, reach 87: ~10 us), then ONE launch
        const band::TilePlan &tp = choice.tile;
        hipLaunchKernelGGL(band::pack_band_kernel, dim3(tp.nblk * tp.Dq4), dim3(64), 0, s, trans, w.tpack, S, tp.hl, tp.hr, tp.Dq,
                           tp.Dq4);
        if (ev) (void)hipEventRecord(ev[3], s);
#define TORBI_BAND_TILE_AS(BPW_, NW_, BG_)                                                                                         \
        {                                                                                                                         \
            e = ensure_dynamic_lds(reinterpret_cast<const void *>(&band::band_tile_kernel<BPW_, NW_, BG_>), (size_t)tp.lds_bytes); \
            if (e != hipSuccess) return e;                                                                                        \
            TORBI_NOTE_KERNEL("band::band_tile_kernel<" #BPW_ ", " #NW_ ", " #BG_ ">");                                           \
            hipLaunchKernelGGL((band::band_tile_kernel<BPW_, NW_, BG_>), dim3(tiles), dim3(64 * tp.waves), (size_t)tp.lds_bytes, s, \
                               grp, tp, w.tpack, init);                                                                          \
        }
#define TORBI_BAND_TILE(BPW_, NW_)                                                                                                \
        if (tp.background != -INFINITY) TORBI_BAND_TILE_AS(BPW_, NW_, true) else TORBI_BAND_TILE_AS(BPW_, NW_, false)
        if (tp.bpw == 1) TORBI_BAND_TILE(1, 12)
        else if (tp.waves == 12) TORBI_BAND_TILE(2, 12)
        else if (tp.bpw == 2) TORBI_BAND_TILE(2, 8)
        else TORBI_BAND_TILE(3, 8)
#undef TORBI_BAND_TILE_AS
#undef TORBI_BAND_TILE
    } else {
        {
            const size_t blocks = std::max<size_t>(1, std::min<size_t>(2048, (most / 16 + 255) / 256));
            hipLaunchKernelGGL(band::clear_exchange_kernel, dim3((unsigned)blocks, n), dim3(256), 0, s, clear);
        }
        if (ev) (void)hipEventRecord(ev[3], s);
        const bool bg = choice.background != -INFINITY;
        e = bg ? ensure_dynamic_lds(reinterpret_cast<const void *>(&band::band_forward_kernel<true>), (size_t)pl.lds_bytes)
               : ensure_dynamic_lds(reinterpret_cast<const void *>(&band::band_forward_kernel<false>), (size_t)pl.lds_bytes);
        if (e != hipSuccess) return e;
        TORBI_NOTE_KERNEL(bg ? "band::band_forward_kernel<true>" : "band::band_forward_kernel<false>");
        // (R > 1: eight dispatch classes of R x ceil(tiles / 8) workgroups each -- band_forward.hpp, membership; a launch
        // never holds more members than one XCD has units for its class: choose_band)
        for (int l = 0; l < nlaunch; ++l) {
            ex.tile0 = l * choice.cap;
            ex.tiles = std::min(tiles, ex.tile0 + choice.cap);
            ex.control = tickets + 8 * l;
            const int here = ex.tiles - ex.tile0;
            const int grid = pl.R > 1 ? 8 * ((here + 7) / 8) * pl.R : here;
            if (bg)
                hipLaunchKernelGGL(band::band_forward_kernel<true>, dim3(grid), dim3(64 * pl.waves), (size_t)pl.lds_bytes, s, grp, ex,
                                   pl, trans, init);
            else
                hipLaunchKernelGGL(band::band_forward_kernel<false>, dim3(grid), dim3(64 * pl.waves), (size_t)pl.lds_bytes, s, grp, ex,
                                   pl, trans, init);
        }
        if (pl.R > 1) {          // does nothing unless a member gave up waiting (band_forward.hpp)
            const size_t lds = 32 * (size_t)S;
            e = ensure_dynamic_lds(reinterpret_cast<const void *>(&band::band_repair_kernel), lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(band::band_repair_kernel, dim3(tiles), dim3(1024), lds, s, grp, ex.failed, trans, init, S, pl.hl,
                               pl.hr, choice.background);
        }
    }
    if (launches) *launches = nlaunch;
    if (ev) (void)hipEventRecord(ev[1], s);
    const int K = backtrace_segments(items);          // (few paths: speculative segments, as behind run_resident)
    int32_t *const arrive = w.base.tile_map;
#define TORBI_BAND_BACKTRACE(NQ_)                                                                                                  \
    if (K > 1) {                                                                                                                   \
        hipLaunchKernelGGL(band::group_segment_band_kernel<NQ_>, dim3(items * K), dim3(64), 0, s, grp, trans, S, pl.hl, pl.hr, K,   \
                           arrive, choice.background);                                                                             \
        hipLaunchKernelGGL(band::group_stitch_band_kernel<NQ_>, dim3(items), dim3(64), 0, s, grp, trans, S, pl.hl, pl.hr, K, arrive, \
                           choice.background);                                                                                     \
    } else {                                                                                                                       \
        hipLaunchKernelGGL(band::group_backtrace_band_kernel<NQ_>, dim3(items), dim3(64), 0, s, grp, trans, S, pl.hl, pl.hr,        \
                           choice.background);                                                                                     \
    }
    if (S <= 512) { TORBI_BAND_BACKTRACE(2) }
    else if (S <= 1536) { TORBI_BAND_BACKTRACE(6) }
    else if (S <= 2048) { TORBI_BAND_BACKTRACE(8) }
    else { TORBI_BAND_BACKTRACE(16) }
#undef TORBI_BAND_BACKTRACE
    {
        const hipError_t ne = nonfinite_end(hb, n, trans, init, S, cus, s);
        if (ne != hipSuccess) return ne;
    }
    if (ev) (void)hipEventRecord(ev[2], s);
    return hipGetLastError();
}

// AUTO takes the band kernel for a promised band when its plan covers the group -- except for shapes a wavefront or a
// workgroup decodes alone (small_states.hpp) and for the handful of sequences of the held-matrix kernel
inline bool band_plan_for(const HostBatch *hb, int n, int S, int hl, int hr, int cus, int path, BandChoice &pl,
                          float background = -INFINITY) {
    if (path != TORBI_HIP_FORWARD_AUTO && path != TORBI_HIP_FORWARD_BAND) return false;
    if (!band_shape(S)) return false;
    int tiles = 0;
    long long items = 0;
    for (int k = 0; k < n; ++k) tiles += band_tiles(hb[k].B), items += hb[k].B;
    if (tiles < 1 || tiles > kMaxGroupTiles) return false;
    if (path == TORBI_HIP_FORWARD_AUTO) {
        if (small::supported(S) || small_block_auto((int)std::min(items, 1ll << 30), S, cus)) return false;
        if (n == 1 && held::supported(hb[0].B, S, cus) && held_auto(hb[0].B, S)) return false;
    }
    return choose_band(S, hl, hr, tiles, cus, pl, background);
}

// one decode on `s`; optional events bracket the forward and backtrace phases (ev[3]: end of the preparation)
hipError_t run_decode(const float *obs, const int32_t *frames, const float *trans, const float *init,
                      int32_t *out, void *workspace, int B, int T, int S, int device, hipStream_t s,
                      hipEvent_t *ev, int *launches, bool reuse, bool collect, int path, unsigned seed_flags = 0u,
                      Route *taken = nullptr) {
    hipError_t e;
    const int cus = cu_count(device);
    Route route = route_for(path, B, S, cus);
    // AUTO keeps away from the held-matrix kernel while another stream of the device is busy (see mark_decode_end)
    if (route == ROUTE_HELD && path == TORBI_HIP_FORWARD_AUTO && other_streams_busy(device, s))
        route = route_for(path, B, S, cus, false);
    // ... and, named or not, from a launch the device cannot hold as a whole (occupancy of this very kernel)
    if (route == ROUTE_HELD && !held_resident(S, device, cus))
        route = route_for(path == TORBI_HIP_FORWARD_HELD ? TORBI_HIP_FORWARD_AUTO : path, B, S, cus, false);
    if (taken) *taken = route;
    if (route == ROUTE_RESIDENT || route == ROUTE_CLUSTER) {
        const HostBatch hb{obs, frames, out, workspace, B, T};
        return run_resident(&hb, 1, trans, init, S, cus, s, ev, launches, reuse, false, route == ROUTE_CLUSTER,
                            few_seeds(seed_flags, route == ROUTE_CLUSTER));
    }
    if (g_preparation) reuse = false;       // (the promise was about the caller's buffer; this route prepares in the workspace)
    if (ev) (void)hipEventRecord(ev[0], s);
    if (ev) (void)hipEventRecord(ev[3], s);
    // NaN / +inf inputs (nonfinite.hpp): the small-state kernels look at the values they produce; the routes below get
    // their observations looked at by a launch of its own (they launch a kernel per timestep anyway)
    const HostBatch alone{obs, frames, out, workspace, B, T};
    e = nonfinite_begin(&alone, 1, trans, init, S, cus, s, route != ROUTE_SMALL, -1, -1, route != ROUTE_SMALL);
    if (e != hipSuccess) return e;
    if (route == ROUTE_SMALL) {             // one launch: recurrence, backtrace and the route record
        // (the byte plane lies where the generic path's trellis does and is never larger: small_states.hpp)
        // (and so does the value-only form's fp32 history: the trellis region itself)
        const Workspace w = carve(workspace, B, T, S);
        int32_t *const record = route_record(workspace, B, T, S, cus);
        e = small::supported(S) ? launch_small(obs, frames, trans, init, w, out, record, B, T, S, s, launches, cus)
                                : launch_block(obs, frames, trans, init, w, out, record, B, T, S, s, launches, cus);
        if (ev) (void)hipEventRecord(ev[1], s);
        if (e == hipSuccess) e = nonfinite_end(&alone, 1, trans, init, S, cus, s);
        if (ev) (void)hipEventRecord(ev[2], s);
        return e;
    }
    e = stamp_route(workspace, B, T, S, cus, route, s);
    if (e != hipSuccess) return e;
    if (route == ROUTE_DENSE) {
        const DenseWorkspace w = carve_dense(workspace, B, T, S, cus);
        // (the route record's words [2], [3]: shader-clock and wall-clock ticks of the last timestep's workgroup 0)
        e = launch_dense_forward(obs, frames, trans, init, w, B, T, S, s, launches, reuse,
                                 reinterpret_cast<unsigned *>(route_record(workspace, B, T, S, cus)) + 2);
        if (ev) (void)hipEventRecord(ev[1], s);
        if (e == hipSuccess)
            e = launch_backtrace_on(w.hist, trans, frames, out, B, T, S, s, w.ranges, w.ranges + 2 * (size_t)S);
    } else if (route == ROUTE_ROWS) {
        TORBI_NOTE_KERNEL("rowscan::step_rows_sorted_kernel");
        const RowsWorkspace w = carve_rows(workspace, B, T, S);
        if (!reuse)       // per-transition preparation: descending rows, prev-states as byte offsets of a 16-item tile row
            hipLaunchKernelGGL(pruned::sort_rows_kernel, dim3(S), dim3(256), sizeof(float) * 2 * (size_t)w.NPOW, s, trans,
                               w.sorted, w.row_range, S, w.SpP, w.NPOW, pruned::kNB * 4);
        {
            const size_t n = (size_t)B * S;
            const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
            hipLaunchKernelGGL(pruned::init_history_kernel, dim3(grid), dim3(256), 0, s, obs, init, w.hist, B, T, S);
        }
        int n = 0;
        for (int t = 1; t < T; ++t, ++n)
            hipLaunchKernelGGL(rowscan::step_rows_sorted_kernel,
                               dim3((S + rowscan::kRowsPerBlock - 1) / rowscan::kRowsPerBlock, B), dim3(256),
                               sizeof(float) * (size_t)S, s, obs, frames, w.sorted, w.hist, w.rowmax, B, T, S, t, w.SpP, 6);
        if (launches) *launches = n;
        e = hipGetLastError();
        if (ev) (void)hipEventRecord(ev[1], s);
        if (e == hipSuccess)
            e = launch_backtrace_sorted(w.hist, w.sorted, w.SpP, pruned::kNB, trans, frames, out, B, T, S, s, w.rowmax);
    } else {
        const Workspace w = carve(workspace, B, T, S);
        TORBI_NOTE_KERNEL(route == ROUTE_HELD ? "held::held_forward_kernel" : "step_rows_kernel / step_rows4_kernel");
        e = route == ROUTE_HELD ? launch_held_forward(obs, frames, trans, init, w, B, T, S, s, launches)
                                : launch_forward(obs, frames, trans, init, w, B, T, S, s, launches);
        if (ev) (void)hipEventRecord(ev[1], s);
        if (e == hipSuccess) e = launch_finalize(frames, w, out, B, T, S, s);
    }
    if (e == hipSuccess) e = nonfinite_end(&alone, 1, trans, init, S, cus, s);
    if (ev) (void)hipEventRecord(ev[2], s);
    return e;
}

struct PhaseEvents {
    hipEvent_t ev[4] = {};
    hipError_t err = hipSuccess;
    PhaseEvents() {
        for (auto &x : ev)
            if (err == hipSuccess) err = hipEventCreate(&x);
    }
    ~PhaseEvents() {
        for (auto &x : ev)
            if (x) (void)hipEventDestroy(x);
    }
    // forward (incl. preparation), argmax + backtrace, preparation alone -- after synchronising on the last event
    hipError_t read(float *phase_ms) {
        hipError_t e = hipEventSynchronize(ev[2]);
        if (e != hipSuccess) return e;
        (void)hipEventElapsedTime(&phase_ms[0], ev[0], ev[1]);
        (void)hipEventElapsedTime(&phase_ms[1], ev[1], ev[2]);
        (void)hipEventElapsedTime(&phase_ms[4], ev[0], ev[3]);
        return hipSuccess;
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int torbi_hip_abi_version(void) { return TORBI_HIP_ABI_VERSION; }

const char *torbi_hip_error_string(int code) {
    switch (code) {
        case TORBI_HIP_OK: return "success";
        case TORBI_HIP_EINVAL: return "invalid argument (null pointer, non-positive dimension or unknown flag)";
        case TORBI_HIP_EWORKSPACE: return "workspace smaller than torbi_hip_workspace_bytes()";
        case TORBI_HIP_ERANGE: return "problem dimensions out of range for this build";
        case TORBI_HIP_ENODEVICE: return "no usable HIP device";
        case TORBI_HIP_EUNSUPPORTED: return "shape not covered by this specialised entry point";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown torbi_hip error";
}

int torbi_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int torbi_hip_compute_units(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return TORBI_HIP_ENODEVICE;
    return cu_count(device);
}

size_t torbi_hip_workspace_bytes(int B, int T, int S) {
    if (B <= 0 || T <= 0 || S <= 0) return 256;
    return need_bytes_any_device(B, T, S);
}

int torbi_hip_last_forward_kernel(char *name_out, size_t capacity) {
    if (!name_out || capacity == 0) return TORBI_HIP_EINVAL;
    snprintf(name_out, capacity, "%s", g_last_kernel);
    return TORBI_HIP_OK;
}

int torbi_hip_set_forward_path(int path) {
    if (path < TORBI_HIP_FORWARD_AUTO || path > TORBI_HIP_FORWARD_BAND) return TORBI_HIP_EINVAL;
    g_forward_path.store(path, std::memory_order_relaxed);
    return TORBI_HIP_OK;
}

int torbi_hip_forward_path(int B, int S) { return torbi_hip_forward_path_on(B, S, 0, 0u); }

int torbi_hip_forward_path_on(int B, int S, int device, unsigned flags) {
    if (B <= 0 || S <= 0 || !flags_ok(flags)) return TORBI_HIP_EINVAL;
    return (int)route_for(requested_path(flags), B, S, cu_count(device));
}

int torbi_hip_viterbi_decode(const float *observation, const int32_t *batch_frames,
                             const float *transition, const float *initial,
                             int32_t *indices_out, void *workspace, size_t workspace_bytes,
                             int B, int T, int S, int device, void *stream) {
    return torbi_hip_viterbi_decode_ex(observation, batch_frames, transition, initial, indices_out, workspace,
                                       workspace_bytes, B, T, S, device, stream, 0u);
}

int torbi_hip_viterbi_decode_ex(const float *observation, const int32_t *batch_frames,
                                const float *transition, const float *initial,
                                int32_t *indices_out, void *workspace, size_t workspace_bytes,
                                int B, int T, int S, int device, void *stream, unsigned flags) {
    if (!flags_ok(flags)) return TORBI_HIP_EINVAL;
    const int rc = check_args(observation, batch_frames, transition, initial, indices_out,
                              workspace, workspace_bytes, B, T, S, device);
    if (rc != TORBI_HIP_OK || B == 0) return rc;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    const hipError_t e = run_decode(observation, batch_frames, transition, initial, indices_out, workspace,
                                    B, T, S, device, static_cast<hipStream_t>(stream), nullptr, nullptr,
                                    (flags & TORBI_HIP_REUSE_TRANSITION) != 0, (flags & TORBI_HIP_COLLECT_STATS) != 0,
                                    requested_path(flags), flags);
    mark_decode_end(device, static_cast<hipStream_t>(stream));
    return (int)e;
}

int torbi_hip_viterbi_decode_batches(const torbi_hip_batch *batches, int count, const float *transition,
                                     const float *initial, int S, int device, void *stream, unsigned flags,
                                     float *phase_ms) {
    if (!flags_ok(flags) || count < 0 || count > TORBI_HIP_MAX_BATCHES || S < 1) return TORBI_HIP_EINVAL;
    if (count == 0) return TORBI_HIP_OK;
    if (!batches || !transition || !initial) return TORBI_HIP_EINVAL;
    if (phase_ms)
        for (int i = 0; i < 6; ++i) phase_ms[i] = 0.0f;
    const int cus = cu_count(device);
    HostBatch hb[TORBI_HIP_MAX_BATCHES];
    int n = 0, tiles = 0;
    for (int k = 0; k < count; ++k) {
        const torbi_hip_batch &b = batches[k];
        const int rc = check_args(b.observation, b.batch_frames, transition, initial, b.indices_out, b.workspace,
                                  b.workspace_bytes, b.B, b.T, S, device);
        if (rc != TORBI_HIP_OK) return rc;
        if (b.B == 0) continue;
        hb[n++] = HostBatch{b.observation, b.batch_frames, b.indices_out, b.workspace, b.B, b.T};
        tiles += tiles_of(b.B, S);
    }
    if (n == 0) return TORBI_HIP_OK;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int path = requested_path(flags);
    const bool reuse = (flags & TORBI_HIP_REUSE_TRANSITION) != 0;
    // ONE time-resident launch for the whole group: named (whole tiles per workgroup, or clusters), or AUTO with enough
    // items -- half the compute units' worth of tiles, or fewer tiles split over clusters (batches of >= 17 items)
    int largest = 0;
    long long items = 0;
    for (int k = 0; k < n; ++k) largest = std::max(largest, hb[k].B), items += hb[k].B;
    const bool split = cluster_members(tiles, S, cus) > 1;
    const bool together = resident_fits(S, tiles) &&
                          (path == TORBI_HIP_FORWARD_RESIDENT || path == TORBI_HIP_FORWARD_CLUSTER ||
                           (path == TORBI_HIP_FORWARD_AUTO && !small::supported(S) && !small_block_auto((int)std::min(items, 1ll << 30), S, cus) &&      // (a wavefront / workgroup per sequence)
                            (2 * tiles > cus || (split && largest > 16))));
    const bool clusters = together && split && path != TORBI_HIP_FORWARD_RESIDENT;
    const bool ascending = (flags & TORBI_HIP_SHORTEST_FIRST) != 0;
    if (phase_ms) {
        PhaseEvents pe;
        if (pe.err != hipSuccess) return (int)pe.err;
        int launches = 0;
        hipError_t e;
        if (together) {
            e = run_resident(hb, n, transition, initial, S, cus, s, pe.ev, &launches, reuse, ascending, clusters,
                             few_seeds(flags, clusters));
            phase_ms[3] = (float)(clusters ? ROUTE_CLUSTER : ROUTE_RESIDENT);
        } else {
            // one batch after the other, each on the path it would take alone; phases of the LAST batch only
            e = hipSuccess;
            // (the reuse promise covers the first batch's workspace only)
            Route taken = route_for(path, hb[n - 1].B, S, cus);
            for (int k = 0; k < n && e == hipSuccess; ++k)
                e = run_decode(hb[k].obs, hb[k].frames, transition, initial, hb[k].out, hb[k].workspace, hb[k].B,
                               hb[k].T, S, device, s, k == n - 1 ? pe.ev : nullptr, &launches, reuse && k == 0, false, path, flags,
                               &taken);
            phase_ms[3] = (float)taken;
        }
        mark_decode_end(device, s);
        if (e == hipSuccess) e = pe.read(phase_ms);
        phase_ms[2] = (float)launches;
        phase_ms[5] = (float)(together ? n : 1);
        return (int)e;
    }
    hipError_t e = hipSuccess;
    if (together)
        e = run_resident(hb, n, transition, initial, S, cus, s, nullptr, nullptr, reuse, ascending, clusters,
                         few_seeds(flags, clusters));
    else
        for (int k = 0; k < n && e == hipSuccess; ++k)
            e = run_decode(hb[k].obs, hb[k].frames, transition, initial, hb[k].out, hb[k].workspace, hb[k].B, hb[k].T, S,
                           device, s, nullptr, nullptr, reuse && k == 0, (flags & TORBI_HIP_COLLECT_STATS) != 0, path, flags);
    mark_decode_end(device, s);
    return (int)e;
}


extern "C++" {
namespace {
// `background_out` null: -inf is what counts as outside; else: the matrix's corner entry, whatever it is
int band_reach_impl(const float *transition, int S, int device, void *stream, int *reach_left_out, int *reach_right_out,
                    float *background_out) {
    if (!transition || S < 1 || !reach_left_out || !reach_right_out) return TORBI_HIP_EINVAL;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a few bytes of device scratch per host thread and device, kept for the life of the thread: hipMalloc / hipFree per call
    // synchronise the whole device (round-5 advisor) and stall other streams' decodes
    thread_local int32_t *scratch[kMaxDevices] = {};
    if (device < 0 || device >= kMaxDevices) return TORBI_HIP_EINVAL;
    hipError_t e = hipSuccess;
    if (!scratch[device]) e = hipMalloc(reinterpret_cast<void **>(&scratch[device]), 4 * sizeof(int32_t));
    if (e != hipSuccess) return (int)e;
    int32_t *const dev = scratch[device];
    int32_t host[4] = {-1, -1, 0, 0};
    hipLaunchKernelGGL(fill_pair_kernel, dim3(1), dim3(1), 0, s, dev, -1, -1);     // (-1, -1: what a matrix without an entry inside leaves)
    hipLaunchKernelGGL(fill_pair_kernel, dim3(1), dim3(1), 0, s, dev + 2, 0, 0);
    hipLaunchKernelGGL(band::band_reach_kernel, dim3(S), dim3(64), 0, s, transition, dev, S, background_out ? 1 : 0);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    *reach_left_out = host[0];
    *reach_right_out = host[1];
    if (background_out) {
        union { int32_t i; float f; } bits;
        bits.i = host[2];
        *background_out = bits.f;
        // a FINITE constant is taken only from matrices whose every other entry lies ABOVE it (a pitch matrix: log(tiny) against
        // at least -10.5 inside the band) and that have such entries: where the band reaches down to the constant, the outputs
        // next to a row's maximum cannot be decided from the maximum alone, and the launch would be decoded again every time
        if (bits.f != -INFINITY && (host[3] != 0 || host[0] < 0)) *reach_left_out = *reach_right_out = S - 1;
    }
    return TORBI_HIP_OK;
}
}  // namespace
}  // extern "C++"

int torbi_hip_band_reach(const float *transition, int S, int device, void *stream, int *reach_left_out, int *reach_right_out) {
    return band_reach_impl(transition, S, device, stream, reach_left_out, reach_right_out, nullptr);
}

int torbi_hip_band_reach_over(const float *transition, int S, int device, void *stream, int *reach_left_out, int *reach_right_out,
                              float *background_out) {
    if (!background_out) return TORBI_HIP_EINVAL;
    return band_reach_impl(transition, S, device, stream, reach_left_out, reach_right_out, background_out);
}

int torbi_hip_band_members_over(int items, int S, int reach_left, int reach_right, float background, int device) {
    if (items < 1 || S < 1 || reach_left < 0 || reach_right < 0) return TORBI_HIP_EINVAL;
    BandChoice c;
    if (!band_shape(S) || !choose_band(S, reach_left, reach_right, band_tiles(items), cu_count(device), c, background)) return 0;
    return c.pl.R;
}

int torbi_hip_band_members(int items, int S, int reach_left, int reach_right, int device) {
    return torbi_hip_band_members_over(items, S, reach_left, reach_right, -INFINITY, device);
}

int torbi_hip_viterbi_decode_banded(const torbi_hip_batch *batches, int count, const float *transition, const float *initial,
                                    int S, int reach_left, int reach_right, int device, void *stream, unsigned flags,
                                    float *phase_ms) {
    return torbi_hip_viterbi_decode_banded_over(batches, count, transition, initial, S, reach_left, reach_right, -INFINITY, device,
                                                stream, flags, phase_ms);
}

int torbi_hip_viterbi_decode_banded_over(const torbi_hip_batch *batches, int count, const float *transition, const float *initial,
                                         int S, int reach_left, int reach_right, float background, int device, void *stream,
                                         unsigned flags, float *phase_ms) {
    if (!flags_ok(flags) || count < 0 || count > TORBI_HIP_MAX_BATCHES || S < 1 || reach_left < 0 || reach_right < 0)
        return TORBI_HIP_EINVAL;
    if (count == 0) return TORBI_HIP_OK;
    if (!batches || !transition || !initial) return TORBI_HIP_EINVAL;
    const int cus = cu_count(device);
    HostBatch hb[TORBI_HIP_MAX_BATCHES];
    int n = 0;
    for (int k = 0; k < count; ++k) {
        const torbi_hip_batch &b = batches[k];
        const int rc = check_args(b.observation, b.batch_frames, transition, initial, b.indices_out, b.workspace,
                                  b.workspace_bytes, b.B, b.T, S, device);
        if (rc != TORBI_HIP_OK) return rc;
        if (b.B == 0) continue;
        hb[n++] = HostBatch{b.observation, b.batch_frames, b.indices_out, b.workspace, b.B, b.T};
    }
    BandChoice pl;
    const int path = requested_path(flags);
    bool vec = (reinterpret_cast<uintptr_t>(transition) & 15) == 0;       // (16-byte reads of matrix rows and observation rows)
    for (int k = 0; k < n; ++k) vec = vec && (reinterpret_cast<uintptr_t>(hb[k].obs) & 15) == 0;
    if (n == 0 || !vec || !band_plan_for(hb, n, S, reach_left, reach_right, cus, path, pl, background)) {
        // not a shape of the band kernel: whatever the plain entry point does with it (BAND named: as AUTO)
        unsigned f = flags;
        if (path == TORBI_HIP_FORWARD_BAND) f = (flags & ~(7u << 4)) | TORBI_HIP_PATH_FLAG(TORBI_HIP_FORWARD_AUTO);
        return torbi_hip_viterbi_decode_batches(batches, count, transition, initial, S, device, stream, f, phase_ms);
    }
    if (phase_ms)
        for (int i = 0; i < 6; ++i) phase_ms[i] = 0.0f;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool ascending = (flags & TORBI_HIP_SHORTEST_FIRST) != 0;
    hipError_t e;
    if (phase_ms) {
        PhaseEvents pe;
        if (pe.err != hipSuccess) return (int)pe.err;
        int launches = 0;
        e = run_band(hb, n, transition, initial, S, pl, cus, s, pe.ev, &launches, ascending);
        mark_decode_end(device, s);
        if (e == hipSuccess) e = pe.read(phase_ms);
        phase_ms[2] = (float)launches;
        phase_ms[3] = (float)ROUTE_BAND;
        phase_ms[5] = (float)n;
        return (int)e;
    }
    e = run_band(hb, n, transition, initial, S, pl, cus, s, nullptr, nullptr, ascending);
    mark_decode_end(device, s);
    return (int)e;
}

size_t torbi_hip_preparation_bytes(int S) { return S > 0 ? preparation_bytes(S) : 256; }

int torbi_hip_viterbi_decode_batches_prepared(const torbi_hip_batch *batches, int count, const float *transition,
                                              const float *initial, int S, int device, void *stream, unsigned flags,
                                              float *phase_ms, void *preparation, size_t preparation_bytes_given,
                                              int *filled) {
    if (filled) *filled = 0;
    if (preparation && (S < 1 || preparation_bytes_given < preparation_bytes(S) || (reinterpret_cast<uintptr_t>(preparation) & 255)))
        return TORBI_HIP_EWORKSPACE;
    struct Scope {
        Scope(void *p, size_t n, bool valid) { g_preparation = p; g_preparation_bytes = n; g_preparation_valid = valid; }
        ~Scope() { g_preparation = nullptr; g_preparation_bytes = 0; g_preparation_valid = false; }
    } scope(preparation, preparation_bytes_given, preparation && (flags & TORBI_HIP_REUSE_TRANSITION));
    const int rc = torbi_hip_viterbi_decode_batches(batches, count, transition, initial, S, device, stream, flags, phase_ms);
    if (filled && rc == TORBI_HIP_OK) *filled = g_preparation_valid ? 1 : 0;
    return rc;
}

int torbi_hip_scan_stats(const void *workspace, size_t workspace_bytes, int B, int T, int S,
                         unsigned *stats_out, int device, void *stream, unsigned flags) {
    if (B < 1 || T < 1 || S < 1 || !workspace || !stats_out || !flags_ok(flags)) return TORBI_HIP_EINVAL;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    if (workspace_bytes < need_bytes(B, T, S, cu_count(device))) return TORBI_HIP_EWORKSPACE;
    // which statistics: decided on the device by the route the last decode with this workspace took
    const int cus = cu_count(device);
    const unsigned *resident_stats = resident::supported(S) ? carve_resident(const_cast<void *>(workspace), B, T, S, cus).stats : nullptr;
    const unsigned *held_control = carve(const_cast<void *>(workspace), B, T, S).control;
    if (!resident_stats && !held_control) return TORBI_HIP_EUNSUPPORTED;
    hipLaunchKernelGGL(gather_stats_kernel, dim3(1), dim3(2 * pruned::kStatSlots), 0, static_cast<hipStream_t>(stream),
                       route_record(workspace, B, T, S, cus), resident_stats ? resident_stats : held_control, held_control,
                       stats_out);
    return (int)hipGetLastError();
}

extern "C++" {
namespace {
template <bool PROBS>
int decode_uniform_as(const float *observation, const int32_t *batch_frames, float log_transition, const float *initial,
                      int32_t *indices_out, int B, int T, int S, int device, void *stream) {
    if (B < 0 || T < 1 || S < 1) return TORBI_HIP_EINVAL;
    if (B == 0) return TORBI_HIP_OK;
    if (!observation || !batch_frames || !initial || !indices_out) return TORBI_HIP_EINVAL;
    if (S % 4 != 0 || S > 4096 || (reinterpret_cast<uintptr_t>(observation) & 15) ||
        (reinterpret_cast<uintptr_t>(initial) & 15))
        return TORBI_HIP_EUNSUPPORTED;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a wave per observation row, the reductions off the dependent chain (uniform_decode.hpp); R rows per wave and chunk:
    // 1, 2 and 3 run alike (0.27 ms at 512 x 500 x 1440), 4 spills
    // at most one workgroup per compute unit: the rows in flight per item are the only parallelism there is -- 16 waves
    // per workgroup (8 above 2048 states: 128 registers a lane do not hold two rows there).  tools/uniform_probe.py, ms per
    // 500 x 1440 decode, 16 / 4 waves: 1 item 0.081 / 0.148, 128: 0.107 / 0.160, 256: 0.141 / 0.206; two 8-wave workgroups
    // per unit for 257..512 items run like the 4-wave ones
    const bool few = B <= cu_count(device);
#define TORBI_UNIFORM_ROWS(NQW_, R_)                                                                     \
    if (S <= 256 * NQW_) {                                                                               \
        constexpr int NWF = NQW_ <= 8 ? 16 : 8;                                                          \
        if (few)                                                                                         \
            hipLaunchKernelGGL((uniform::uniform_rows_kernel<NQW_, 1, PROBS, NWF>), dim3(B), dim3(64 * NWF), 0, s, \
                               observation, batch_frames, initial, log_transition, indices_out, B, T, S); \
        else                                                                                             \
            hipLaunchKernelGGL((uniform::uniform_rows_kernel<NQW_, R_, PROBS>), dim3(B), dim3(256), 0, s, \
                               observation, batch_frames, initial, log_transition, indices_out, B, T, S); \
        hipLaunchKernelGGL((uniform::uniform_repair_kernel<PROBS>), dim3(B), dim3(256), 0, s, observation, batch_frames, \
                           initial, log_transition, indices_out, B, T, S);     /* (items that read a NaN / +inf) */ \
        mark_decode_end(device, s);                                                                      \
        return (int)hipGetLastError();                                                                   \
    }
    TORBI_UNIFORM_ROWS(1, 2)
    TORBI_UNIFORM_ROWS(2, 2)
    TORBI_UNIFORM_ROWS(4, 2)
    TORBI_UNIFORM_ROWS(6, 2)
    TORBI_UNIFORM_ROWS(8, 2)
    TORBI_UNIFORM_ROWS(12, 1)
    TORBI_UNIFORM_ROWS(16, 1)
#undef TORBI_UNIFORM_ROWS
    return TORBI_HIP_EUNSUPPORTED;
}
}  // namespace
}  // extern "C++"

int torbi_hip_viterbi_decode_uniform(const float *observation, const int32_t *batch_frames,
                                     float log_transition, const float *initial,
                                     int32_t *indices_out, int B, int T, int S, int device,
                                     void *stream) {
    return decode_uniform_as<false>(observation, batch_frames, log_transition, initial, indices_out, B, T, S, device, stream);
}

int torbi_hip_viterbi_decode_uniform_probabilities(const float *probabilities, const int32_t *batch_frames,
                                                   float log_transition, const float *initial, int32_t *indices_out,
                                                   int B, int T, int S, int device, void *stream) {
    return decode_uniform_as<true>(probabilities, batch_frames, log_transition, initial, indices_out, B, T, S, device, stream);
}

int torbi_hip_viterbi_decode_profiled(const float *observation, const int32_t *batch_frames,
                                      const float *transition, const float *initial,
                                      int32_t *indices_out, void *workspace,
                                      size_t workspace_bytes, int B, int T, int S, int device,
                                      void *stream, unsigned flags, float *phase_ms) {
    if (!phase_ms || !flags_ok(flags)) return TORBI_HIP_EINVAL;
    const torbi_hip_batch one{observation, batch_frames, indices_out, workspace, workspace_bytes, B, T};
    // a single batch through the batches entry: same routing as torbi_hip_viterbi_decode_ex for this shape (AUTO
    // there counts the 16-item tiles of the group, which for one batch is what route_for() does)
    unsigned f = flags;
    if (((flags >> 4) & 7u) == 0) f |= TORBI_HIP_PATH_FLAG(default_path());
    return torbi_hip_viterbi_decode_batches(&one, 1, transition, initial, S, device, stream, f, phase_ms);
}

int torbi_hip_read_posterior(const void *workspace, size_t workspace_bytes,
                             const int32_t *batch_frames, float *posterior_out, int B, int T,
                             int S, int device, void *stream, unsigned flags) {
    if (B < 0 || T < 1 || S < 1 || !flags_ok(flags)) return TORBI_HIP_EINVAL;
    if (B == 0) return TORBI_HIP_OK;
    if (!workspace || !batch_frames || !posterior_out) return TORBI_HIP_EINVAL;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    if (workspace_bytes < need_bytes(B, T, S, cu_count(device))) return TORBI_HIP_EWORKSPACE;
    const size_t n = (size_t)B * S;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    const Workspace w = carve(const_cast<void *>(workspace), B, T, S);
    hipLaunchKernelGGL(gather_final_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       route_record(workspace, B, T, S, cu_count(device)), static_cast<const float *>(workspace), w.post[0],
                       w.post[1], batch_frames, posterior_out, B, T, S);
    return (int)hipGetLastError();
}

int torbi_hip_epsilon_clamp(float *x, uint64_t count, int device, void *stream) {
    if (count == 0) return TORBI_HIP_OK;
    if (!x || (reinterpret_cast<uintptr_t>(x) & 15)) return TORBI_HIP_EINVAL;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    const uint64_t blocks = (count / 4 + 255) / 256 + 1;
    const int grid = (int)(blocks < 16384 ? blocks : 16384);
    hipLaunchKernelGGL(epsilon_clamp_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       count);
    return (int)hipGetLastError();
}

int torbi_hip_log_epsilon_clamp(const float *probabilities, float *out, uint64_t count, int device, void *stream) {
    if (count == 0) return TORBI_HIP_OK;
    if (!probabilities || !out || ((reinterpret_cast<uintptr_t>(probabilities) | reinterpret_cast<uintptr_t>(out)) & 15))
        return TORBI_HIP_EINVAL;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    const uint64_t blocks = (count / 4 + 255) / 256 + 1;
    const int grid = (int)(blocks < 16384 ? blocks : 16384);
    hipLaunchKernelGGL(log_epsilon_clamp_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), probabilities,
                       out, count);
    return (int)hipGetLastError();
}

int torbi_hip_fill_synthetic(float *dst, uint64_t count, uint64_t start, int stream_id, int seed,
                             int device, void *stream) {
    if (count == 0) return TORBI_HIP_OK;
    if (!dst) return TORBI_HIP_EINVAL;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return (int)guard.err;
    const uint64_t key =
        ((uint64_t)(uint32_t)stream_id + (uint64_t)(uint32_t)seed * 1000003ull) * 0x9E3779B97F4A7C15ull;
    const uint64_t blocks = (count + 255) / 256;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(fill_synthetic_kernel, dim3(grid), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dst, count, start, key);
    return (int)hipGetLastError();
}

}  // extern "C"

#ifdef BAND_STAMP
// instrumentation build only (tools/band_stamps.py); not part of include/torbi_hip.h
extern "C" int torbi_hip_debug_band_phases(unsigned long long *host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(band::g_phase), count * sizeof(unsigned long long));
}
#endif
#ifdef RESIDENT_STAMP
// instrumentation build only (tools/resident_stamps.py); not part of include/torbi_hip.h
extern "C" int torbi_hip_debug_phases(unsigned long long *host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(resident::g_phase), count * sizeof(unsigned long long));
}
extern "C" int torbi_hip_debug_wgtime(unsigned long long *host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(resident::g_wgtime), count * sizeof(unsigned long long));
}
#endif
