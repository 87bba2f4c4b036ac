// resident_forward.hpp -- the exact pruned forward recurrence (pruned_forward.hpp) with the TIME LOOP INSIDE the
// kernel: one workgroup owns 16 batch items x ALL next-states for every timestep of those items.
//
// Why: one launch per timestep of this recurrence (step_pruned_kernel, rounds 1-3, since removed) spent 6.9 of its
// 19.3 us scanning; the rest is what a kernel boundary costs when every state tile needs the whole previous posterior --
// re-staging the 92 KB posterior tile from L2 in each of the 8 state tiles (23.6 MB per launch), merging per-tile top
// lists, waiting for the slowest of 256 workgroups (HISTORY.md 4.3).  Here the 16 items' posterior rows never leave the
// LDS: a timestep is
//     barrier -> every wave scans its row groups against the LDS tile (outputs stay in registers, go to hist)
//     -> barrier -> the outputs overwrite the tile -> next timestep
// with no inter-workgroup traffic at all (batch items are independent: viterbi.cpp:65, viterbi.cu:58).  The price
// is parallelism: a 512-item batch is only 32 workgroups, so this path is for MANY items in flight -- several
// batches decoded by one launch (torbi_hip_viterbi_decode_batches: grid = sum of the batches' tiles), or one
// batch of >= 2048 items.  Ragged batches cost nothing extra: every workgroup loops to the longest of ITS 16
// items, not to the batch maximum.
//
// Arithmetic, bound and seeds are those of pruned_forward.hpp (same sorted/arranged lists, same transposed
// matrix), so posterior rows -- and the indices lazy_backtrace.hpp recomputes from them -- are bit-identical.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>

#include "pruned_forward.hpp"
#include "lazy_backtrace.hpp"
#include "nonfinite.hpp"

namespace resident {

using pruned::kBlk;
using pruned::kR;
using pruned::kTop;
using pruned::ListBlock;
using pruned::group_bcast;
using pruned::load_list_block;

constexpr int kNI = 16;            // batch items per tile up to kMaxS16 states (4 item groups of 4 per next-state: a lane
                                   // quad, 16 next-states per wave pass); 8 above (2 lanes per next-state, 32 per pass):
                                   // the posterior tile [S][items] fp32 has to fit the 160 KB LDS
constexpr int kMaxBatches = 16;    // batches one launch can carry
constexpr int kRowGroup = 16;      // next-states per wave pass of the 16-item form
// items per tile / next-states per wave pass for S states
#ifndef TORBI_TILE16_MAX_S
#define TORBI_TILE16_MAX_S pruned::kMaxS16      // (experiments: -DTORBI_TILE16_MAX_S=0 runs every shape on 8-item tiles)
#endif
__host__ __device__ inline int tile_items(int S) { return S <= TORBI_TILE16_MAX_S ? kNI : kNI / 2; }
__host__ __device__ inline int pass_rows(int S) { return 64 / (tile_items(S) / 4); }

typedef unsigned long long u64;

struct Batch {
    const float *obs;        // (B,T,S)
    const int32_t *frames;   // (B)
    float *hist;             // (B,T,S) posterior history of this batch
    int32_t *out;            // (B,T) decoded indices (backtrace)
    const int32_t *order;    // (B) items by descending length: tile k owns items order[16k .. 16k+15]
    float *rowmax;           // (B,T) largest entry of every posterior row (the gather form of the backtrace bounds with it)
    int B, T;
    int tile0;               // first workgroup of this batch in the launch
    int item0;               // first item of this batch in the launch-wide item numbering (backtrace grid)
    int32_t *alarm;          // raised (= Group::serial) by a kernel that produced a NaN / +inf posterior value (nonfinite.hpp)
};

struct Group {
    Batch batch[kMaxBatches];
    int n;
    const int32_t *tile_map;   // [tiles of the group] workgroup -> (batch << 20 | tile of that batch), longest first
    unsigned *stats;           // [128] scan statistics: [0] list blocks walked, [64] wave passes counted (sampled)
    const unsigned *only;      // null, or [tiles]: workgroup w decodes its tile only where only[w] != 0 -- the repair launch
                               // behind a cluster launch (Cluster::failed): tiles whose cluster gave up waiting, whole again
    int serial;                // of this decode: what an alarm is raised with
};

// CLUSTER form: R workgroups (a cluster) share one 16-item tile.  Every member keeps the WHOLE posterior tile in its
// LDS but scans only its share of the next-states; after a timestep the members exchange their slices of the new row
// through `xchg` (write-through stores of self-validating data, a hint per member and timestep: cluster_slot_bytes below;
// MI355X_MICROARCH.md "transport-variants") -- no kernel boundary, no grid-wide barrier, the clusters drift freely.
// That puts a single 512-item batch (32 tiles) on 256 compute units.  Membership is by ARRIVAL (a ticket drawn at
// kernel entry) within a DISPATCH CLASS (workgroups b, b + 8, b + 16, ...): nothing depends on dispatch order or placement,
// and a cluster whose last members have not been dispatched yet only waits -- every cluster that is complete runs to its
// end and frees its compute units.  The GPU places a class on one XCD (observed, not promised); the members compare notes
// at kernel entry, and a cluster that finds itself on ONE XCD exchanges through that XCD's L2 -- plain stores that stay in
// it, L1-bypassing loads that hit it -- instead of write-through stores and reads across the fabric (round 5; no faster
// on the benchmark: profiles/r05_cluster_exchange.txt).
#ifndef RESIDENT_MAX_R
#define RESIDENT_MAX_R 16
#endif
constexpr int kMaxR = RESIDENT_MAX_R;    // members per cluster
constexpr int kMaxTop = 4;               // list entries per item a member publishes (>= KR + 1 of every instantiation)
// The exchange is SELF-VALIDATING (round 5): a slot starts out as kAbsentBits in every word -- a NaN no fp32 addition of
// the recurrence produces -- and a consumer takes a 16-byte piece of a posterior row when none of its four words is that
// pattern any more; the partial top lists travel as {key low, timestep, key high, timestep}.  So a producer neither waits
// for the acknowledgement of its slice stores nor publishes anything behind them, and a consumer asks for the pieces the
// moment its own rows are in its tile: the load that finds them IS the hand-off, a piece that is not there yet is asked for
// again after a pause.  Two of the three dependent trips of a timestep (store acknowledgement, flag) are gone from the
// critical path.  Four slots by timestep: row t goes to slot t % 4 and, behind the workgroup barrier of
// timestep t, every wave resets the slot row t - 2 went to -- which every member finished reading before it sent row t - 1,
// which this member has taken -- two timesteps before the next use ((t + 2) % 4), with that timestep's observation loads
// (issued behind the reset, awaited before any output exists: vmcnt counts in order) between the two.  The assumption
// (gfx9: ONE vmcnt for loads and stores, decremented in issue order) is pinned by tools/cluster_soak.py: every decode
// against the dense route's, 105 842 decodes on two streams without a mismatch (profiles/r06_soak_stress.txt).
// (An input that makes the recurrence produce this very NaN -- NaNs are out of contract -- runs into the bounded wait and is
// decoded again by the repair launch, like any cluster that cannot complete.)
constexpr unsigned kAbsentBits = 0x7fd5a5a5u;
constexpr int kSlots = 4;
// bytes of one exchange slot: a posterior row of the tile + the members' partial top lists
__host__ __device__ inline size_t cluster_slot_bytes(int S) {
    return ((size_t)S + 3) / 4 * 4 * kNI * sizeof(float) +
           (size_t)kMaxR * kNI * kMaxTop * 2 * sizeof(u64);
}
struct Cluster {
    float *xchg;           // [tiles][kSlots] slots by timestep: the members' slices of the newest posterior rows
                           // [S4][16] floats, then their partial top lists [kMaxR][16 * kMaxTop] tagged 64-bit keys
                           // (every word kAbsentBits before the launch: absent_kernel)
    unsigned *where;       // [tiles][kMaxR] the XCD each member runs on, + 1 (zeroed before the launch)
    int tiles;             // tiles of the launch (the grid is padded to whole dispatch classes: 8 x ceil(tiles / 8) x R)
    unsigned *control;     // [0 .. 7] tickets drawn per dispatch class (zeroed before the launch); give-ups are counted
                           // in Group::stats[127]
    unsigned *failed;      // [tiles] set by a member that gave up waiting for the others (zeroed before the launch): the
                           // tile's history is incomplete and the launch that follows decodes it again, whole
    int R;
    unsigned long long wait_ticks;     // how long a member waits for the others' flags (100 MHz ticks; 0: not at all)
    int spread;            // 1 (R % 8 == 0): member m of EVERY tile runs on XCD m / (R / 8) -- the rows' sorted lists, which the
                           // members of all tiles walk, stay in that XCD's L2; the exchange then crosses XCDs (write-through)
};
// Cluster::wait_ticks, default (ticks of the 100 MHz wall clock): a quarter of a second -- members that have not been
// dispatched yet because another stream's launch holds the compute units arrive within tens of milliseconds; a cluster
// that can never complete (a device shared with a process that never leaves) must not hang the call.
// TORBI_HIP_CLUSTER_WAIT_US overrides (0 = give up at the first poll that fails: the tests of the repair launch).
constexpr unsigned long long kClusterWaitTicks = 25000000ull;

inline bool supported(int S) { return S >= 64 && S <= pruned::kMaxS; }

// dynamic LDS: posterior tile [S4][16] + running top lists (64-bit keys) + decoded top lists + frame counts + items
// + 4 control words (cluster ticket, gave-up flag)
inline size_t lds_bytes(int S, int ktop = kTop) {
    const size_t S4 = ((size_t)S + 3) / 4 * 4, ni = (size_t)tile_items(S);
    return sizeof(float) * ni * S4 + sizeof(u64) * kNI * ktop + (sizeof(float) + sizeof(int)) * kNI * ktop +
           2 * sizeof(int) * kNI + 4 * sizeof(int);
}

// row groups (16 next-states) member m of R scans: [m * nrg / R, (m + 1) * nrg / R)
__host__ __device__ inline int cluster_first_group(int m, int nrg, int R) { return (int)((long long)m * nrg / R); }

// once per decode: order[rank] = item, items ranked by descending (clamped) length, ties by item number.  A tile of
// 16 consecutive ranks then loops to the longest of 16 items of SIMILAR length -- a ragged batch costs recurrence
// steps for its valid frames only (collate pads every item of a 512-batch to the batch maximum: reference
// torbi/data/collate.py:24-31) -- and the longest tiles are dispatched first.  Results do not depend on the order.
// grid = (ceil(max B / 256), batches), block = 256; O(B^2) compares up to kMaxOrdered items; larger batches are ranked by
// a counting sort over the lengths (order_large_* below: items of equal length in arrival order of their atomics --
// which of two equally long items lands in which tile changes nothing, neither results nor work).
constexpr int kMaxOrdered = 8192;
struct OrderJob { const int32_t *frames; int32_t *order; int B, T, tile0; int32_t *hist; int32_t *route_record; int route; };
struct OrderJobs { OrderJob job[kMaxBatches]; int ascending; int n; int tiles; int32_t *tile_map; unsigned *stats;
                   unsigned *flags; int nflags;         // cluster form: the flag / ticket words to zero (else nflags = 0)
                   int ni; };                           // items per tile (16, or 8 above kMaxS16 states)

__global__ __launch_bounds__(256) void order_items_kernel(OrderJobs jobs) {
    const OrderJob &jb = jobs.job[blockIdx.y];
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int B = jb.B, T = jb.T;
    if (b == 0 && jb.route_record) *jb.route_record = jb.route;      // the route this batch's decode takes (torbi_hip.hip)
    if (b >= B) return;
    if (B > kMaxOrdered) return;                     // order_large_* rank this batch
    int f = jb.frames[b];
    f = f < 1 ? 1 : (f > T ? T : f);
    int rank = 0;
    for (int o = 0; o < B; ++o) {
        int g = jb.frames[o];
        g = g < 1 ? 1 : (g > T ? T : g);
        rank += (g > f) || (g == f && o < b);
    }
    jb.order[jobs.ascending ? B - 1 - rank : rank] = b;
}

// Batches above kMaxOrdered items: hist[f] = items of (clamped) length f (zeroed by the host), turned into the first rank
// of every length by one block, then every item takes the next rank of its length.
__global__ __launch_bounds__(256) void order_large_count_kernel(OrderJob jb) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= jb.B) return;
    int f = jb.frames[b];
    f = f < 1 ? 1 : (f > jb.T ? jb.T : f);
    atomicAdd(&jb.hist[f], 1);
}

__global__ __launch_bounds__(1024) void order_large_scan_kernel(OrderJob jb, int ascending) {
    // hist[f] <- number of items that rank before the items of length f (longer ones; shorter ones when ascending)
    __shared__ int part[1024];
    const int T = jb.T, tid = threadIdx.x;
    const int per = (T + 1024) / 1024;                       // lengths 1..T in 1024 contiguous chunks, in rank order
    auto length_at = [&](int k) { return ascending ? 1 + k : T - k; };      // k-th length in rank order
    int sum = 0;
    for (int k = tid * per; k < (tid + 1) * per && k < T; ++k) sum += jb.hist[length_at(k)];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    int run = part[tid];
    for (int k = tid * per; k < (tid + 1) * per && k < T; ++k) {
        const int f = length_at(k), v = jb.hist[f];
        jb.hist[f] = run;
        run += v;
    }
}

__global__ __launch_bounds__(256) void order_large_place_kernel(OrderJob jb) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= jb.B) return;
    int f = jb.frames[b];
    f = f < 1 ? 1 : (f > jb.T ? jb.T : f);
    jb.order[atomicAdd(&jb.hist[f], 1)] = b;
}

// ... and the group's TILES by descending length of their longest item: workgroup w decodes tile_map[w].  The GPU hands
// workgroup i to shader engine i mod 32 and waits, in launch order, for a free CU THERE (tools/dispatch_probe.hip,
// tools/resident_wgtime.py): with batch-major numbering the rank-c tiles of all batches -- all of one length -- meet on
// one engine, the engine of the longest tiles is the one the first workgroup beyond the 256th waits for, and nothing
// behind it starts until those finish (measured: 28.3 ms for a ragged 16-batch group that has 16.5 ms of work per CU).
// Ranked across the whole group, every engine holds an even spread of lengths and frees a CU early.
// One thread per tile; grid = ceil(tiles / 256); O(tiles^2) compares.  Runs after order_items_kernel.
__global__ __launch_bounds__(256) void order_tiles_kernel(OrderJobs jobs) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w < 128) jobs.stats[w] = 0u;                 // the forward launch that follows accumulates into them
    for (int k = w; k < jobs.nflags; k += gridDim.x * 256) jobs.flags[k] = 0u;     // cluster flags and tickets start at zero
    if (w >= jobs.tiles) return;
    auto tile_length = [&](int k, int j) {
        const OrderJob &jb = jobs.job[k];
        // the longest item of tile j: its first in descending order, its last (within the batch) in ascending order
        const int last = jobs.ni * j + jobs.ni - 1 < jb.B ? jobs.ni * j + jobs.ni - 1 : jb.B - 1;
        int f = jb.frames[jb.order[jobs.ascending ? last : jobs.ni * j]];
        return f < 1 ? 1 : (f > jb.T ? jb.T : f);
    };
    int k = 0;
    for (int q = 1; q < jobs.n; ++q)
        if (w >= jobs.job[q].tile0) k = q;
    const int j = w - jobs.job[k].tile0;
    const int mine = tile_length(k, j);
    int rank = 0;
    for (int q = 0; q < jobs.n; ++q) {
        const int nt = (jobs.job[q].B + jobs.ni - 1) / jobs.ni;
        for (int t = 0; t < nt; ++t) {
            const int other = tile_length(q, t);
            const int ow = jobs.job[q].tile0 + t;
            rank += (other > mine) || (other == mine && ow < w);
        }
    }
    jobs.tile_map[jobs.ascending ? jobs.tiles - 1 - rank : rank] = (k << 20) | j;
}

// every word of the exchange slots a cluster launch will use := kAbsentBits (grid-stride, 16 bytes per thread and step)
__global__ __launch_bounds__(256) void absent_kernel(uint4 *__restrict__ slots, size_t granules) {
    const uint4 a = {kAbsentBits, kAbsentBits, kAbsentBits, kAbsentBits};
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < granules; g += (size_t)gridDim.x * 256) slots[g] = a;
}

// order-preserving 64-bit key of (value, state): larger value first, then the lower state
__device__ __forceinline__ u64 top_key(float v, int state) {
    unsigned u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    return ((u64)u << 32) | (unsigned)(0x7fffffff - state);
}

// insert into a kTop-entry list kept in descending order by a cascade of LDS atomic maxima: the displaced key
// moves one rank down, so every rank ends with the maximum of what passed through it
// (a key that is already in the list is dropped: the members of a cluster see each other's keys more than once)
template <int KTOP = kTop>
__device__ __forceinline__ void top_insert(u64 *list, u64 x) {
    if (x <= list[KTOP - 1]) return;
#pragma unroll
    for (int r = 0; r < KTOP; ++r) {
        if (x != 0ull) {
            const u64 old = atomicMax(&list[r], x);
            x = old < x ? old : (old == x ? 0ull : x);
        }
    }
}

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
// 16-byte write-through (sc1) store / L1-bypassing (sc1) load at byte `offset` of a buffer: the payload side of the
// cluster hand-off (a plain store would stay in the producer XCD's L2, a plain load could hit a stale L1 line)
__device__ __forceinline__ void store_through(__amdgpu_buffer_rsrc_t buffer, int offset, float4 v) {
    v4u x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, buffer, offset, 0, 16);
}
__device__ __forceinline__ void store_plain(__amdgpu_buffer_rsrc_t buffer, int offset, float4 v) {
    v4u x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, buffer, offset, 0, 0);
}
__device__ __forceinline__ float4 load_through(__amdgpu_buffer_rsrc_t buffer, int offset) {
    const v4u x = __builtin_amdgcn_raw_buffer_load_b128(buffer, offset, 0, 16);
    return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// build-time ablations for tools/variants_probe.py (timing only, results are wrong): bit 0 fixed scan depth of 11 blocks
// (no termination test), 1 no posterior reads from the LDS, 2 no quad broadcasts, 3 no history stores, 4 no top-list
// inserts, 5 no observation loads, 6 no seeds, 7 no exchange stores (cluster form).  RESIDENT_EXTRA_VALU=n adds n independent v_add_f32 per entry pair (results
// unchanged): how much of the run time is the vector instruction stream
#ifndef RESIDENT_ABL
#define RESIDENT_ABL 0
#endif
// Cluster form: the pause (64-cycle units) between two attempts of a wave at slices that were not there yet
#ifndef CLUSTER_POLL_SLEEP
#define CLUSTER_POLL_SLEEP 4
#endif
#ifndef CLUSTER_ROUND
#define CLUSTER_ROUND 8                // 16-byte pieces of the other members' slices a thread asks for at once
#endif
#ifndef RESIDENT_EXTRA_VALU
#define RESIDENT_EXTRA_VALU 0
#endif

#ifdef RESIDENT_STAMP
// build-time instrumentation (tools/resident_stamps.py): per-wave cycle sums of the phases of a timestep
constexpr int kPhases = 12;     // 0..6 timestep phases, 7 extra list blocks, 8..11 cluster: drain, flag wait, slices, barrier
__device__ unsigned long long g_phase[1024 * 16 * kPhases];
__device__ unsigned long long g_wgtime[1024 * 4];        // per workgroup: start, end (100 MHz wall clock), steps, -
#define RSTAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); acc[i] += now_ - last; last = now_; }
#define RCOUNT(i, n) acc[i] += (n)
#else
#define RSTAMP(i)
#define RCOUNT(i, n)
#endif

// ---------------------------------------------------------------------------------------
// The whole forward pass of 16 items.  grid = tiles of every batch of the group, block = 64 * KW,
// dynamic LDS = lds_bytes(S).  MAXP = ceil(ceil(S/16) / KW) row-group passes per wave and timestep.
// ---------------------------------------------------------------------------------------
// PIPE: the posterior reads of an entry pair are issued while the previous pair's adds/maxima run (two pairs of
// ds_read_b128 in flight per wave, also across list blocks: the first pair of the next block is read before the
// termination test decides whether it is needed) -- with one timestep per launch the scan was a third of the
// kernel and this bought nothing (tools/prune_proto5.hip); here the scan IS the kernel.
// KR: seeds per item (explicit candidates; thr = the (KR+1)-th largest posterior).  CLUSTER: see struct Cluster.
// NI: items per tile (16; 8 for kMaxS16 < S <= kMaxS: G = NI / 4 lanes per next-state, 64 / G next-states per wave pass).
// REPAIR: the launch behind a cluster launch (Cluster::failed): a workgroup decodes its tile only where Group::only says so
// (an instance of its own, so that its -- normally empty -- dispatches do not count as the forward kernel's in a profile).
template <int KW, int MAXP, bool PIPE, int KR = kR, bool CLUSTER = false, int NI = 16, bool REPAIR = false>
__global__ __launch_bounds__(64 * KW) void resident_forward_kernel(Group grp, Cluster clu, const float *__restrict__ tt,
                                                                   const float2 *__restrict__ sorted,
                                                                   const float *__restrict__ initial, int S, int SpP) {
    constexpr int kNI = NI, G = NI / 4, EPL = kBlk / G, kRowGroup = 64 / G;     // (shadow the namespace-wide 16-item values)
    constexpr int kR = KR, kTop = KR + 1;
    static_assert(NI == 16 || NI == 8, "tiles of 16 or 8 items");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S4 = (S + 3) / 4 * 4;
    u64 *top = reinterpret_cast<u64 *>(lds + (size_t)kNI * S4);       // [16][kTop] this timestep's largest outputs
    float *mtopv = reinterpret_cast<float *>(top + kNI * kTop);       // [16][kTop] previous timestep's, decoded
    int *mtopi = reinterpret_cast<int *>(mtopv + kNI * kTop);         // their states, as offsets into tt (state * S)
    int *sframes = mtopi + kNI * kTop;                                // [16] frames per item (0 past the batch)
    int *sitem = sframes + kNI;                                       // [16] item numbers (a valid one past the batch)
    int *smisc = sitem + kNI;                                         // [0] cluster ticket, [1] gave up waiting

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (workgroup 0 leaves the shader-clock and the 100 MHz wall-clock ticks of its run in stats[120], [121]: the clock
    // delivered under this kernel's load -- what the vector ALU's ceiling is priced at)
    const unsigned long long clock_0 = clock64(), wall_0 = wall_clock64();
#ifdef RESIDENT_STAMP
    const unsigned long long wg_start = wall_clock64();
#endif

    // which tile of which batch this workgroup decodes; CLUSTER: by arrival -- ticket / R is the cluster (= tile),
    // ticket % R the member
    int cid = blockIdx.x, member = 0;
    const int R = CLUSTER ? clu.R : 1;
    if constexpr (REPAIR) {
        if (!grp.only || grp.only[blockIdx.x] == 0u) return;     // (this tile's cluster completed)
    }
    bool local = false;                  // CLUSTER: every member of this cluster runs on one XCD
    if constexpr (CLUSTER) {
        const int cls = blockIdx.x & 7;
        if (tid == 0) {
            smisc[0] = (int)__hip_atomic_fetch_add(clu.control + cls, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            smisc[1] = 0;
        }
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(smisc[0]);
        if (clu.spread) {
            const int per = R / 8;           // members of a tile per XCD
            member = cls * per + ticket % per;
            cid = ticket / per;
        } else {
            member = ticket % R;
            cid = (ticket / R) * 8 + cls;
        }
        if (cid >= clu.tiles) return;        // (the grid is padded to whole classes)
        // where do the R members run?  Each says so (write-through), all read all R answers (bounded wait; a cluster that
        // cannot complete in time exchanges write-through and gives up at its first hand-off as before)
        if (wave == 0) {
            unsigned *const where = clu.where + (size_t)cid * kMaxR;
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;      // HW_REG_XCC_ID
            if (lane == 0) __hip_atomic_store(where + member, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long since = 0ull;
            unsigned seen = xcc + 1u;
            bool complete = true;
            for (;;) {
                if (lane < R) seen = __hip_atomic_load(where + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(seen != 0u)) break;
                const unsigned long long now = wall_clock64();
                if (since == 0ull) since = now;
                if (now - since >= clu.wait_ticks) { complete = false; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            const bool same = complete && __all(seen == xcc + 1u);
            if (lane == 0) smisc[2] = same ? 1 : 0;
        }
        __syncthreads();
#ifndef CLUSTER_LOCAL_EXCHANGE
#define CLUSTER_LOCAL_EXCHANGE 1             // (0: write-through exchange whatever the placement -- experiments)
#endif
        local = CLUSTER_LOCAL_EXCHANGE && smisc[2] != 0;
    }
    const int code = grp.tile_map[cid];
    const Batch &bat = grp.batch[code >> 20];
    const float *__restrict__ obs = bat.obs;
    float *__restrict__ hist = bat.hist;
    const int B = bat.B, T = bat.T;
    const int b0 = (code & 0xfffff) * kNI;
    bool odd = false;                  // a NaN / +inf posterior value was produced (nonfinite.hpp)

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
    if (tid < kNI * kTop) top[tid] = 0ull;
    __syncthreads();
    int fmax = 0;
#pragma unroll
    for (int it = 0; it < kNI; ++it) fmax = max(fmax, sframes[it]);

    // t = 0: posterior row 0 = obs[b,0,:] + initial (viterbi.cpp:72-76) into the tile, the history and the top lists
    for (int item = 0; item < kNI; ++item) {
        const int b = sitem[item];
        const bool valid = b0 + item < B;
        const float *src = obs + (size_t)b * T * S;
        float *dst = hist + (size_t)b * T * S;
        for (int i = tid; i < S; i += 64 * KW) {
            const float v = src[i] + initial[i];
            odd = odd || nonfinite::odd(v);
            lds[i * kNI + item] = v;
            if (valid) dst[i] = v;
            top_insert<kTop>(top + item * kTop, top_key(v, i));
        }
    }

    // lane = next-state jl of the wave's 16 x item group g; quads of lanes -> next-states so that every
    // ds_read_b128 lane group holds an aligned row quad (what arrange_blocks_kernel<4> keeps conflict-poor)
    // (8-item tiles: lane pairs -> next-states, aligned groups of eight rows per lane group: arrange_blocks_kernel<8>)
    const int g = lane & (G - 1);
    const int jl = G == 4 ? (int)((0xFBAE9DC873261540ull >> (4 * (lane >> 2))) & 15)
                          : (int)((0xFE7654DC32BA9810ull >> (4 * ((lane >> 1) & 15))) & 15) + (lane & 32) / 2;
    const char *ptile = reinterpret_cast<const char *>(lds) + 16 * g;
    const int nrg = (S + kRowGroup - 1) / kRowGroup;
    // this workgroup's row groups: all of them, or its share as member of a cluster
    const int rg_lo = CLUSTER ? cluster_first_group(member, nrg, R) : 0;
    const int rg_hi = CLUSTER ? cluster_first_group(member + 1, nrg, R) : nrg;
    // CLUSTER: bytes of a posterior row of the tile and of a whole slot of the exchange buffer (slot = 2 * cid + parity)
    const unsigned xrow = (unsigned)S4 * kNI * (unsigned)sizeof(float);
    const unsigned xbytes = (unsigned)cluster_slot_bytes(S);
    static_assert(kTop <= kMaxTop, "a member publishes at most kMaxTop list entries per item");
    // items of this lane: tile items 4g .. 4g+3; items past the batch read a valid one's observations and store nothing
    int ib[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) ib[it] = sitem[4 * g + it];

    float pend[MAXP][4];
    // CLUSTER: the 16-byte pieces of the other members' slices (every row but this member's own), thread by thread
    const int own_lo = kRowGroup * rg_lo, own_hi = kRowGroup * rg_hi < S ? kRowGroup * rg_hi : S;
    const int pieces = CLUSTER ? (S - (own_hi - own_lo)) * G : 0;
    auto place = [&](int c) {          // byte offset of piece c (tile and slot alike)
        int row = c / G;
        row = row < own_lo ? row : row + (own_hi - own_lo);
        return (row * kNI + 4 * (c % G)) * 4;
    };
    // A cluster member whose waves scan ONE row group per timestep (MAXP == 1) meets the same sorted rows every
    // timestep: their first two list blocks stay in registers, and the next timestep's observations are requested
    // before the wait for the other members (everything a pass needs that does not depend on the exchange).
#ifndef RESIDENT_FIXED8
#define RESIDENT_FIXED8 1
#endif
#ifndef RESIDENT_FIXED16
#define RESIDENT_FIXED16 1
#endif
    constexpr bool FIXED = CLUSTER && MAXP == 1 && (NI == 16 ? RESIDENT_FIXED16 : RESIDENT_FIXED8);
    ListBlock<EPL> head0, head1;
    float obnext[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FIXED) {
        const int rg = rg_lo + wave;
        const int jj = kRowGroup * rg + jl;
        const int jr = (rg < rg_hi && jj < S) ? jj : S - 1;
        const float2 *row = sorted + (size_t)jr * SpP + EPL * g;
        load_list_block(head0, row, 0);
        load_list_block(head1, row, kBlk);
        if (fmax > 1) {
#pragma unroll
            for (int it = 0; it < 4; ++it) obnext[it] = obs[((size_t)ib[it] * T + 1) * S + jr];
        }
    }
    // scan statistics for adaptive path selection (every 16th timestep): how many 16-entry list blocks a wave pass
    // walks.  The benchmark needs 10.6 of the 90 a row holds; near 90 nothing is being pruned and the dense kernel wins.
    unsigned stat_blocks = 0, stat_passes = 0;

    // publish the largest entries of the row the tile holds (decoded; states as offsets into tt) and empty the
    // running lists for the next row's outputs
    // (`row`: the timestep of the row the tile holds; its maximum goes to bat.rowmax for the backtrace, once per cluster)
    auto publish_top = [&](int row) {
        if (tid < kNI * kTop) {
            const u64 key = top[tid];
            unsigned u = (unsigned)(key >> 32);
            u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
            const float value = key ? __uint_as_float(u) : -INFINITY;
            mtopv[tid] = value;
            mtopi[tid] = key ? (0x7fffffff - (int)(unsigned)key) * S : 0;
            top[tid] = 0ull;
            const int item = tid / kTop;
            if (tid == item * kTop && member == 0 && b0 + item < B && row < sframes[item])
                bat.rowmax[(size_t)sitem[item] * T + row] = value;
        }
    };
    __syncthreads();
    publish_top(0);
#ifdef RESIDENT_STAMP
    unsigned long long acc[kPhases] = {};
    unsigned long long last = __builtin_readcyclecounter();
#endif

    for (int t = 1; t < fmax; ++t) {
        __syncthreads();      // tile = posterior row t-1, mtop = its largest entries, `top` is empty
        RSTAMP(0);

        // seeds and bound of this lane's four items: the kR largest posteriors are explicit candidates, the
        // (kR+1)-th bounds every other one.  Items that have ended (t >= frames) get thr = -inf: their bound
        // never asks for another list block.
        float seedv[4][kR ? kR : 1], thr[4];          // (kR = 0: no seeds at all, thr = the largest posterior)
        int seedo[4][kR ? kR : 1];
        bool live[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = 4 * g + it;
            live[it] = t < sframes[item];
            if constexpr (kTop == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(mtopv + item * kTop);
                const int4 o = *reinterpret_cast<const int4 *>(mtopi + item * kTop);
                const float vv[4] = {v.x, v.y, v.z, v.w};
                const int oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int r = 0; r < kR; ++r) { seedv[it][r] = vv[r]; seedo[it][r] = oo[r]; }
                thr[it] = live[it] ? vv[kR] : -INFINITY;
            } else {
#pragma unroll
                for (int r = 0; r < kR; ++r) { seedv[it][r] = mtopv[item * kTop + r]; seedo[it][r] = mtopi[item * kTop + r]; }
                thr[it] = live[it] ? mtopv[item * kTop + kR] : -INFINITY;
            }
        }
        RSTAMP(1);

        // (an opaque zero keeps the row-group addresses of all MAXP passes from being hoisted out of the time loop,
        // where they would cost ~10 registers per pass)
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int rg = rg_lo + wave + KW * p + opaque;  // wave-uniform
            if (rg < rg_hi) {
                const int jj = kRowGroup * rg + jl;
                const bool jv = jj < S;
                const int jr = jv ? jj : S - 1;
                const float2 *row = sorted + (size_t)jr * SpP + EPL * g;
                ListBlock<EPL> cur, nxt;
                float ob[4];
                if constexpr (FIXED) {
                    cur = head0;
                    nxt = head1;
#pragma unroll
                    for (int it = 0; it < 4; ++it) ob[it] = obnext[it];
                } else {
                    load_list_block(cur, row, 0);
                    load_list_block(nxt, row, kBlk);
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        ob[it] = (RESIDENT_ABL & 32) ? 0.5f * it : obs[((size_t)ib[it] * T + t) * S + jr];
                }
                float seedt[4][kR ? kR : 1];
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int r = 0; r < kR; ++r)
                        seedt[it][r] = (RESIDENT_ABL & 64) ? -1.0f : tt[(unsigned)(seedo[it][r] + jr)];   // trans[jr][i_r]

                float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#if RESIDENT_EXTRA_VALU
                float dummy[4] = {0.f, 0.f, 0.f, 0.f};
#endif
                struct PairData { float4 p0, p1; float t0, t1; };
                // entry pair H (0..7) of a block: owner lane O = H / 2 of the quad, its e[H % 2]
                auto issue = [&](auto Hc, const ListBlock<EPL> &blk, PairData &d) {
                    constexpr int H = decltype(Hc)::value, O = H / (EPL / 2), E = H % (EPL / 2);
                    float t0, t1;
                    int o0, o1;
                    if (RESIDENT_ABL & 4) {
                        t0 = blk.e[E].x; t1 = blk.e[E].z; o0 = __float_as_int(blk.e[E].y); o1 = __float_as_int(blk.e[E].w);
                    } else {
                        t0 = group_bcast<G, O>(blk.e[E].x); t1 = group_bcast<G, O>(blk.e[E].z);
                        o0 = group_bcast<G, O>(__float_as_int(blk.e[E].y));
                        o1 = group_bcast<G, O>(__float_as_int(blk.e[E].w));
                    }
                    asm volatile("" : "+v"(t0), "+v"(t1));     // keep the broadcasts out of the adds (half-rate DPP adds)
                    d.t0 = t0; d.t1 = t1;
                    if (RESIDENT_ABL & 2) {
                        d.p0 = make_float4(__int_as_float(o0), t1, t0, t1);
                        d.p1 = make_float4(__int_as_float(o1), t0, t1, t0);
                    } else {
                        d.p0 = *reinterpret_cast<const float4 *>(ptile + o0);
                        d.p1 = *reinterpret_cast<const float4 *>(ptile + o1);
                    }
                };
                auto math = [&](const PairData &d) {
#if RESIDENT_EXTRA_VALU
                    // probe: extra independent full-rate adds per entry pair (is the vector ALU the limit?)
#pragma unroll
                    for (int x = 0; x < RESIDENT_EXTRA_VALU; ++x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(dummy[x & 3]) : "v"(d.t0));
#endif
                    best[0] = fmaxf(fmaxf(best[0], d.t0 + d.p0.x), d.t1 + d.p1.x);
                    best[1] = fmaxf(fmaxf(best[1], d.t0 + d.p0.y), d.t1 + d.p1.y);
                    best[2] = fmaxf(fmaxf(best[2], d.t0 + d.p0.z), d.t1 + d.p1.z);
                    best[3] = fmaxf(fmaxf(best[3], d.t0 + d.p0.w), d.t1 + d.p1.w);
                };
                PairData ahead;            // PIPE: pair 0 of the block `consume` is about to be called on
                // all 8 pairs of `blk`; PIPE: pair 0 is already in `ahead`, and pair 0 of `after` is left there
                auto consume = [&](const ListBlock<EPL> &blk, const ListBlock<EPL> &after) {
                    if (PIPE) {
                        PairData other;
#define TORBI_STAGE(H_, CUR_, NXT_)                                                   \
                        issue(std::integral_constant<int, H_ + 1>(), blk, NXT_);      \
                        __builtin_amdgcn_sched_barrier(0);                            \
                        math(CUR_);                                                   \
                        __builtin_amdgcn_sched_barrier(0);
                        TORBI_STAGE(0, ahead, other)
                        TORBI_STAGE(1, other, ahead)
                        TORBI_STAGE(2, ahead, other)
                        TORBI_STAGE(3, other, ahead)
                        TORBI_STAGE(4, ahead, other)
                        TORBI_STAGE(5, other, ahead)
                        TORBI_STAGE(6, ahead, other)
#undef TORBI_STAGE
                        issue(std::integral_constant<int, 0>(), after, ahead);
                        __builtin_amdgcn_sched_barrier(0);
                        math(other);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        PairData d;
                        issue(std::integral_constant<int, 0>(), blk, d); math(d);
                        issue(std::integral_constant<int, 1>(), blk, d); math(d);
                        issue(std::integral_constant<int, 2>(), blk, d); math(d);
                        issue(std::integral_constant<int, 3>(), blk, d); math(d);
                        issue(std::integral_constant<int, 4>(), blk, d); math(d);
                        issue(std::integral_constant<int, 5>(), blk, d); math(d);
                        issue(std::integral_constant<int, 6>(), blk, d); math(d);
                        issue(std::integral_constant<int, 7>(), blk, d); math(d);
                    }
                };
                if (PIPE) issue(std::integral_constant<int, 0>(), cur, ahead);
                int nblk = 1;                              // wave-uniform
                consume(cur, nxt);
                load_list_block(cur, row, 2 * kBlk);
                auto fold_seeds = [&]() {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int r = 0; r < kR; ++r) best[it] = fmaxf(best[it], seedv[it][r] + seedt[it][r]);
                };
                auto more = [&](const ListBlock<EPL> &blk) {
                    if (RESIDENT_ABL & 1) return nblk < 11;
                    const float tn = group_bcast<G, 0>(blk.e[0].x);
                    return (bool)__any(jv && ((tn + thr[0] > best[0]) | (tn + thr[1] > best[1]) | (tn + thr[2] > best[2]) |
                                              (tn + thr[3] > best[3])));
                };
                const int Sp = (S + 15) / 16 * 16;
                fold_seeds();
                RSTAMP(2);
                for (int kk = kBlk; kk < Sp; kk += 2 * kBlk) {
                    if (!more(nxt)) break;
                    RCOUNT(7, 1);
                    ++nblk;
                    consume(nxt, cur);
                    load_list_block(nxt, row, kk + 2 * kBlk);
                    if (!more(cur)) break;
                    RCOUNT(7, 1);
                    ++nblk;
                    consume(cur, nxt);
                    load_list_block(cur, row, kk + 3 * kBlk);
                }
                if ((t & 15) == 1) { stat_blocks += (unsigned)nblk; stat_passes += 1u; }
                RSTAMP(3);
                // the four items' list thresholds are read together (one LDS round trip, not four in a row)
                u64 last4[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) last4[it] = top[(4 * g + it) * kTop + kTop - 1];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
#if RESIDENT_EXTRA_VALU
                    if (dummy[it] == 12345.678f) best[it] = 0.f;
#endif
                    const float o = ob[it] + best[it];                     // post'[j] = obs[t,j] + max
                    odd = odd || nonfinite::odd(o);
                    pend[p][it] = o;
                    if (jv && live[it] && !(RESIDENT_ABL & 8)) hist[((size_t)ib[it] * T + t) * S + jr] = o;
                    const u64 key = top_key(o, jr);
                    if (jv && key > last4[it] && !(RESIDENT_ABL & 16)) top_insert<kTop>(top + (4 * g + it) * kTop, key);
                }
                RSTAMP(4);
            }
        }
        // CLUSTER: every row's 16 outputs go to the other members as one write-through (sc1) 64-byte row, 16 bytes per lane
        // of the quad, in the layout of the LDS tile.  All of a wave's rows are stored HERE, behind its last pass, not pass
        // by pass: a write-through store stays in the wave's memory queue until memory has acknowledged it, and the history
        // stores and list loads of the next pass queued up behind it (14 of 50 us per timestep with two passes per wave).
        // Nobody waits for the acknowledgement: the data validates itself at the consumer (cluster_slot_bytes).
        if constexpr (CLUSTER) {
            if (t + 1 < fmax && !(RESIDENT_ABL & 128)) {
                const __amdgpu_buffer_rsrc_t xdst = buffer_of(
                    reinterpret_cast<const char *>(clu.xchg) + (size_t)(kSlots * cid + (t & (kSlots - 1))) * xbytes, xrow);
#pragma unroll
                for (int p = 0; p < MAXP; ++p) {
                    const int rg = rg_lo + wave + KW * p + opaque;
                    const int jj = kRowGroup * rg + jl;
                    if (rg < rg_hi && jj < S) {
                        const float4 row4 = make_float4(pend[p][0], pend[p][1], pend[p][2], pend[p][3]);
                        if (local) store_plain(xdst, (jj * kNI + 4 * g) * 4, row4);       // (stays in this XCD's L2)
                        else store_through(xdst, (jj * kNI + 4 * g) * 4, row4);
                    }
                }
            }
            RSTAMP(8);
        }
        __syncthreads();      // every wave is done reading the tile and mtop, every output is in `top`
        RSTAMP(5);
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int rg = rg_lo + wave + KW * p + opaque;
            const int jj = kRowGroup * rg + jl;
            if (rg < rg_hi && jj < S)
                *reinterpret_cast<float4 *>(lds + (size_t)jj * kNI + 4 * g) =
                    make_float4(pend[p][0], pend[p][1], pend[p][2], pend[p][3]);
        }
        if constexpr (CLUSTER) {
            // this wave's rows of the slot row t - 2 went to: absent again, two timesteps before row t + 2 arrives there
            if (t + 1 < fmax) {
                const __amdgpu_buffer_rsrc_t xold = buffer_of(
                    reinterpret_cast<const char *>(clu.xchg) + (size_t)(kSlots * cid + ((t + 2) & (kSlots - 1))) * xbytes, xrow);
                unsigned absent_bits = kAbsentBits;
                asm volatile("" : "+v"(absent_bits));      // (made here: four registers held across the loop would be spilled)
                const float absent = __uint_as_float(absent_bits);
                const float4 none = make_float4(absent, absent, absent, absent);
#pragma unroll
                for (int p = 0; p < MAXP; ++p) {
                    const int rg = rg_lo + wave + KW * p + opaque;
                    const int jj = kRowGroup * rg + jl;
                    if (rg < rg_hi && jj < S) {
                        if (local) store_plain(xold, (jj * kNI + 4 * g) * 4, none);
                        else store_through(xold, (jj * kNI + 4 * g) * 4, none);
                    }
                }
                asm volatile("" ::: "memory");          // (ahead of the observation loads below, in program order)
            }
        }
        if constexpr (FIXED) {
            if (t + 1 < fmax) {
                const int jj = kRowGroup * (rg_lo + wave) + jl;
                const int jr = (rg_lo + wave < rg_hi && jj < S) ? jj : S - 1;
#pragma unroll
                for (int it = 0; it < 4; ++it) obnext[it] = obs[((size_t)ib[it] * T + t + 1) * S + jr];
            }
        }
        if constexpr (CLUSTER) {
            if (t + 1 < fmax) {
                const __amdgpu_buffer_rsrc_t xsrc = buffer_of(
                    reinterpret_cast<const char *>(clu.xchg) + (size_t)(kSlots * cid + (t & (kSlots - 1))) * xbytes, xbytes);
                // (1) this member's partial top lists, tagged with the timestep, behind the posterior row of the slot
                if (wave == 0 && lane < kNI * kTop) {
                    const u64 k = top[lane];
                    v4u x = {(unsigned)k, (unsigned)t, (unsigned)(k >> 32), (unsigned)t};
                    const int at = (int)xrow + (member * kNI * kMaxTop + lane) * 16;
                    if (local) __builtin_amdgcn_raw_buffer_store_b128(x, xsrc, at, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(x, xsrc, at, 0, 16);
                }
                RSTAMP(6);
                // (2) their slices of row t -> the tile (every 16-byte piece except this member's own rows), their partial
                // top lists -> this workgroup's lists: asked for AT ONCE, all loads of a thread in flight together (eight
                // pieces a round: 7 at 1440 states and eight members); a piece that still reads absent, a key with another
                // timestep's tag, is asked for again after a pause -- the load that finds the data is the hand-off, there
                // is no flag to wait for first (a hint flag per member, polled by one wave ahead of the loads, cost a
                // dependent trip: 14.2 against 13.5 us per timestep at 512 x 1440, 26.3 against 22.0 at 128 x 4096).
                // Bounded: a wave that has waited Cluster::wait_ticks gives up for its workgroup
                {
                    const int kpair = kNI * kTop / 2;          // keys: two per thread (R * 16 * kTop / 2 <= 512 threads)
                    const int km = tid / kpair, ks = tid - km * kpair;
                    bool want_keys = km < R && km != member;
                    const bool keyed = want_keys;
                    v4u key0 = {0u, 0u, 0u, 0u}, key1 = {0u, 0u, 0u, 0u};
                    const int kat = (int)xrow + (km * kNI * kMaxTop + 2 * ks) * 16;
                    constexpr int kRound = CLUSTER_ROUND;
                    unsigned spins = 0;
                    unsigned long long since = 0ull;
                    bool gave_up = false;
                    for (int first = 0; first < pieces && !gave_up; first += kRound * 64 * KW) {
                        float4 got[kRound];
                        unsigned missing = 0u;
#pragma unroll
                        for (int u = 0; u < kRound; ++u) {
                            missing |= (first + tid + u * 64 * KW < pieces ? 1u : 0u) << u;
                            got[u] = make_float4(0.f, 0.f, 0.f, 0.f);       // (a piece that is never asked for counts as there)
                        }
                        for (;;) {
#pragma unroll
                            for (int u = 0; u < kRound; ++u)
                                if (missing >> u & 1u) got[u] = load_through(xsrc, place(first + tid + u * 64 * KW));
                            if (want_keys) {
                                key0 = __builtin_amdgcn_raw_buffer_load_b128(xsrc, kat, 0, 16);
                                key1 = __builtin_amdgcn_raw_buffer_load_b128(xsrc, kat + 16, 0, 16);
                            }
#pragma unroll
                            for (int u = 0; u < kRound; ++u) {
                                const bool there = __float_as_uint(got[u].x) != kAbsentBits && __float_as_uint(got[u].y) != kAbsentBits &&
                                                   __float_as_uint(got[u].z) != kAbsentBits && __float_as_uint(got[u].w) != kAbsentBits;
                                if (there) missing &= ~(1u << u);
                            }
                            if (want_keys && key0.y == (unsigned)t && key0.w == (unsigned)t && key1.y == (unsigned)t &&
                                key1.w == (unsigned)t)
                                want_keys = false;
                            if (!__any(missing != 0u || want_keys)) break;
                            RCOUNT(9, 1);
                            __builtin_amdgcn_s_sleep(CLUSTER_POLL_SLEEP);
                            if ((spins++ & 15u) == 0u) {
                                const unsigned long long now = wall_clock64();
                                if (since == 0ull) since = now;
                                if (now - since >= clu.wait_ticks) {
                                    if (lane == 0) smisc[1] = 1;
                                    gave_up = true;
                                    break;
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < kRound; ++u) {
                            const int c = first + tid + u * 64 * KW;
                            if (c < pieces) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(lds) + place(c)) = got[u];
                        }
                    }
                    if (keyed && !gave_up) {
                        const u64 k0 = ((u64)key0.z << 32) | key0.x, k1 = ((u64)key1.z << 32) | key1.x;
                        if (k0) top_insert<kTop>(top + (2 * ks / kTop) * kTop, k0);
                        if (k1) top_insert<kTop>(top + ((2 * ks + 1) / kTop) * kTop, k1);
                    }
                }
                RSTAMP(10);
                __syncthreads();      // the tile holds row t, `top` its largest entries
                RSTAMP(11);
                if (smisc[1]) break;  // (uniform: read behind the barrier)
            }
        }
        publish_top(t);
        RSTAMP(6);
    }
    if constexpr (CLUSTER) {
        if (tid == 0 && smisc[1]) {
            atomicAdd(&grp.stats[127], 1u);     // workgroups that gave up waiting (0 on any sane run)
            clu.failed[cid] = 1u;               // ... and their tile is decoded again by the launch that follows
        }
    }
    if (lane == 0 && stat_passes) {
        atomicAdd(&grp.stats[0], stat_blocks);
        atomicAdd(&grp.stats[64], stat_passes);
    }
    nonfinite::raise(odd, bat.alarm, grp.serial);
    if (!REPAIR && blockIdx.x == 0 && tid == 0) {
        grp.stats[120] = (unsigned)((clock64() - clock_0) >> 4);         // (in units of 16 ticks: a launch may run for seconds)
        grp.stats[121] = (unsigned)((wall_clock64() - wall_0) >> 4);
    }
#ifdef RESIDENT_STAMP
    if (lane == 0 && blockIdx.x < 1024)
        for (int i = 0; i < kPhases; ++i) g_phase[((size_t)blockIdx.x * 16 + wave) * kPhases + i] = acc[i];
    if (tid == 0 && blockIdx.x < 1024) {
        g_wgtime[4 * blockIdx.x] = wg_start;
        g_wgtime[4 * blockIdx.x + 1] = wall_clock64();
        g_wgtime[4 * blockIdx.x + 2] = (unsigned long long)fmax;
    }
#endif
}

// ---------------------------------------------------------------------------------------
// final state, tail fill and lazy backtrace (lazy_backtrace.hpp) for every item of every batch of the group in
// ONE launch: grid = items of the whole group, one wave per item.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int batch_of_item(const Group &grp, int item) {
    int k = 0;
#pragma unroll 1
    for (int q = 1; q < grp.n; ++q)
        if (item >= grp.batch[q].item0) k = q;
    return k;
}

template <int NQ>
__global__ __launch_bounds__(64) void group_backtrace_prefetch_kernel(Group grp, const float *__restrict__ trans, int S) {
    const Batch &bat = grp.batch[batch_of_item(grp, blockIdx.x)];
    const int b = (int)blockIdx.x - bat.item0;
    lazy::backtrace_prefetch_item<NQ>(bat.hist + (size_t)b * bat.T * S, trans, bat.frames[b], bat.out + (size_t)b * bat.T,
                                      bat.T, S, threadIdx.x);
}

template <int NQ>
__global__ __launch_bounds__(64) void group_backtrace_sorted_kernel(Group grp, const float2 *__restrict__ sorted, int SpP,
                                                                    int S) {
    extern __shared__ __attribute__((aligned(16))) float hrow_lds[];
    const Batch &bat = grp.batch[batch_of_item(grp, blockIdx.x)];
    const int b = (int)blockIdx.x - bat.item0;
    // (list offsets = state * bytes of a tile row: 64 with 16-item tiles, 32 with 8-item tiles)
    lazy::backtrace_sorted_item<NQ>(bat.hist + (size_t)b * bat.T * S, sorted, SpP, tile_items(S) == kNI ? 6 : 5,
                                    bat.frames[b], bat.out + (size_t)b * bat.T, bat.T, S, threadIdx.x, hrow_lds);
}

// the same with the posteriors gathered from the history where the list points (no row staging): for launches with many
// waves per compute unit, which are bound by the bytes they move, not by the latency of a step (lazy_backtrace.hpp)
template <int NQ>
__global__ __launch_bounds__(64) void group_backtrace_gather_kernel(Group grp, const float2 *__restrict__ sorted, int SpP, int S) {
    const Batch &bat = grp.batch[batch_of_item(grp, blockIdx.x)];
    const int b = (int)blockIdx.x - bat.item0;
    lazy::backtrace_gather_item<NQ>(bat.hist + (size_t)b * bat.T * S, bat.rowmax + (size_t)b * bat.T, sorted, SpP,
                                    tile_items(S) == kNI ? 6 : 5, bat.frames[b], bat.out + (size_t)b * bat.T, bat.T, S,
                                    threadIdx.x);
}

// ... in K speculative segments per item (lazy_backtrace.hpp, chase_segment / stitch_segments): grid = items x K, then items
template <int NQ>
__global__ __launch_bounds__(64) void group_segment_gather_kernel(Group grp, const float2 *__restrict__ sorted, int SpP, int S,
                                                                  int K, int32_t *__restrict__ arrive) {
    const int item = (int)blockIdx.x / K, seg = (int)blockIdx.x - item * K;
    const Batch &bat = grp.batch[batch_of_item(grp, item)];
    const int b = item - bat.item0;
    const lazy::GatherWalker<NQ> w{bat.hist + (size_t)b * bat.T * S, bat.rowmax + (size_t)b * bat.T, sorted, SpP,
                                   tile_items(S) == kNI ? 6 : 5, S, (int)threadIdx.x};
    lazy::chase_segment(w, bat.frames[b], bat.T, K, seg, bat.out + (size_t)b * bat.T, arrive + (size_t)item * K, threadIdx.x);
}
template <int NQ>
__global__ __launch_bounds__(64) void group_stitch_gather_kernel(Group grp, const float2 *__restrict__ sorted, int SpP, int S,
                                                                 int K, const int32_t *__restrict__ arrive) {
    const int item = blockIdx.x;
    const Batch &bat = grp.batch[batch_of_item(grp, item)];
    const int b = item - bat.item0;
    const lazy::GatherWalker<NQ> w{bat.hist + (size_t)b * bat.T * S, bat.rowmax + (size_t)b * bat.T, sorted, SpP,
                                   tile_items(S) == kNI ? 6 : 5, S, (int)threadIdx.x};
    lazy::stitch_segments(w, bat.frames[b], bat.T, K, bat.out + (size_t)b * bat.T, arrive + (size_t)item * K, threadIdx.x,
                          grp.stats + 122);
}

template <int VEC>
__global__ __launch_bounds__(64) void group_backtrace_kernel(Group grp, const float *__restrict__ trans, int S) {
    const Batch &bat = grp.batch[batch_of_item(grp, blockIdx.x)];
    const int b = (int)blockIdx.x - bat.item0;
    lazy::backtrace_item<VEC>(bat.hist + (size_t)b * bat.T * S, trans, bat.frames[b], bat.out + (size_t)b * bat.T, bat.T, S,
                              threadIdx.x);
}

}  // namespace resident
