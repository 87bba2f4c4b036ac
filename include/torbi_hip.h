/*
 * torbi_hip.h -- C ABI of libtorbi_hip.so, the MI355X (gfx950) batched Viterbi decoder.
 *
 * This is the drop-in boundary for ONE path of maxrmorrison/torbi: the operator
 *
 *     torbi::viterbi_decode(Tensor observation, Tensor batch_frames,
 *                           Tensor transition, Tensor initial) -> Tensor
 *
 * declared at reference torbi/csrc/ops.cpp:16-18, implemented for CUDA at
 * torbi/csrc/cuda/viterbi.cu:309-362 (viterbi_decode_cuda) and called from
 * torbi/viterbi.py:53.  The entry points below are what a binding for that operator
 * binds; INTEGRATION.md shows the ctypes stub and the torch.library registration.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / ATen types
 *   - all pointers are DEVICE pointers on HIP device `device` unless stated otherwise
 *   - tensors are contiguous row-major (the reference calls .contiguous() itself,
 *     viterbi.cu:325-328)
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     device's default stream), performs no allocation, keeps no pointer after return, and
 *     is re-entrant across devices/streams
 *   - return value: 0 = success, < 0 = argument error (TORBI_HIP_E*), > 0 = a hipError_t
 *     reported by the runtime (launch failure etc.)
 *
 * Result contract (what "same results as the reference" means; reference CPU path,
 * torbi/csrc/viterbi.cpp, line numbers in DESIGN.md):
 *   - transition is indexed [next, prev]; candidate(j,i) = fl(post[i] + transition[j*S+i])
 *   - post'[j] = fl(observation[t,j] + max_i candidate(j,i)); t = 0: fl(obs[0,i]+initial[i])
 *   - backpointer = LOWEST index attaining the maximum; final state = lowest index attaining
 *     the maximum of the last posterior row; every output position t >= batch_frames[b]-1
 *     holds that final state
 *   - decoded indices are bit-identical to the reference CPU operator, -inf, +inf and NaN
 *     inputs included: the reference is deterministic there (a NaN candidate at prev-state 0
 *     is never replaced and a NaN candidate elsewhere never wins, viterbi.cpp:94-100; the
 *     final state is ATen's argmax, the first NaN of the last row, :218), the fast kernels
 *     are not, so every decode looks for NaN / +inf in what it reads and produces and an
 *     item that met one is decoded again on the device in the reference's own order of
 *     evaluation (csrc/nonfinite.hpp: no host involvement, nothing to pay on finite inputs
 *     beyond two small launches).  batch_frames[b] outside [1, T] is clamped on the device
 *     (the reference reads out of bounds for 0, viterbi.cpp:153).
 */
#ifndef TORBI_HIP_H
#define TORBI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TORBI_HIP_ABI_VERSION 15

#define TORBI_HIP_OK 0
#define TORBI_HIP_EINVAL (-1)      /* null pointer / non-positive dimension            */
#define TORBI_HIP_EWORKSPACE (-2)  /* workspace smaller than torbi_hip_workspace_bytes */
#define TORBI_HIP_ERANGE (-3)      /* dimension too large for this build               */
#define TORBI_HIP_ENODEVICE (-4)   /* no usable HIP device / wrong architecture        */
#define TORBI_HIP_EUNSUPPORTED (-5) /* shape not covered by this specialised entry point */

/* Build/ABI version of the loaded library (== TORBI_HIP_ABI_VERSION). */
int torbi_hip_abi_version(void);

/* Human-readable text for a return code of this library (static storage). */
const char *torbi_hip_error_string(int code);

/* Number of visible HIP devices (0 if none or the runtime failed to initialise). */
int torbi_hip_device_count(void);

/* Compute units of `device` as the tiling plans see them (256 on a whole MI355X, fewer on a
 * partitioned one); TORBI_HIP_ENODEVICE for an index that does not exist.  Queried once per device. */
int torbi_hip_compute_units(int device);

/*
 * Bytes of device scratch `torbi_hip_viterbi_decode` needs for a (B,T,S) problem.
 * Replaces the reference's internal at::zeros trellis (B,T,S) int32 + posterior (B,S)
 * fp32 allocations (viterbi.cu:331-336): the CALLER owns the scratch (e.g. a torch uint8
 * tensor from the caching allocator).  Contents need no initialisation.
 */
size_t torbi_hip_workspace_bytes(int B, int T, int S);

/*
 * Forward-recurrence paths (all give identical indices):
 *   SMALL     2 <= S <= 256: the matrix in registers for the whole decode (small_states.hpp).  Up to 64 states one
 *             wavefront per sequence, ONE launch: lane j keeps row j of the matrix, a timestep is the previous row broadcast
 *             through the LDS and S x (add, compare, max, select), byte backpointers walked back by the same wavefront
 *             (many sequences: a value-only form, the first argmax recomputed along the path).  65 .. 256 states one
 *             workgroup of ceil(S / 64)^2 waves per sequence (or per two), value-only: wave (nb, pq) keeps next-states
 *             64 nb .. x a quarter / third / half of the prev-states, the pieces of a row meet through the LDS, posterior
 *             rows go to the history and the backtrace is a launch pair of its own (speculative segments, joined
 *             exactly).  AUTO's choice up to 192 states, and up to 256 while B S^2 <= 12 * 2^16 per compute unit (larger
 *             batches: RESIDENT / CLUSTER, whose pruning then wins).  No path name: a named path that covers the shape
 *             runs instead.
 *   GENERIC   S = 1, S > 4096 with B < 32, or DENSE named for a batch below 32 items (any S): trellis kernels shaped like
 *             the reference's, one launch per timestep
 *   ROWS      B <= 16, 64 <= S <= 4096: the pruned recurrence with one wave per (item, next-state), 64 list
 *             entries per wave step (small_batch_forward.hpp); one launch per timestep.  AUTO takes it for
 *             6 .. 16 items while B S <= 12 * 1440 (beyond that one tile split over sixteen workgroups -- CLUSTER -- is
 *             faster; below six items both it and GENERIC are bound by the gap between launches)
 *   DENSE     value-only (max,+) GEMM, every (prev, next) cell evaluated, one launch per timestep
 *   (PRUNED)  the exact pruned recurrence -- sorted transition rows + per-item top posteriors bound the cells that can
 *             still win, the rest are never touched -- is what ROWS, RESIDENT and CLUSTER run.  Its one-launch-per-timestep
 *             tile form (rounds 1-3, route number 2) was removed in round 4: TORBI_HIP_FORWARD_PRUNED now names ROWS up to
 *             16 items and the time-resident forms above.
 *   RESIDENT  the pruned recurrence with the time loop inside ONE launch: a workgroup owns 16 items (8 above 2048
 *             states) x all states for every timestep, the posterior rows never leave its LDS (64 <= S <= 4096, any B).
 *             One tile per compute unit: the path for many items in flight -- several batches through
 *             torbi_hip_viterbi_decode_batches, or one batch of more than 8 * compute-units items.
 *   CLUSTER   the same kernel with each tile split over R <= 16 workgroups that scan 1/R of the next-states each and
 *             exchange their slices of every new posterior row inside the launch (self-validating 16-byte pieces: no
 *             flag, no store drain; resident_forward.hpp): the path for launches that fill at most half the compute units.  Named for a launch
 *             that fills more, it is RESIDENT.
 *   HELD      B <= 16, S <= 4096: the reference's scan with the time loop inside ONE launch: ceil(S / 8) workgroups
 *             hold 8 rows of the matrix each in registers for the whole launch and pass the posterior rows to each
 *             other through 8-byte {value, timestep} words (held_matrix_forward.hpp).
 *   BAND      (ABI 14; torbi_hip_viterbi_decode_banded only, which is told the band) a transition matrix that is -inf outside a
 *             band j - reach_left <= i <= j + reach_right -- the reference's own pitch model, torbi/evaluate/core.py:24-33 --
 *             with the time loop inside ONE launch: R workgroups share a 16-item tile, each keeps its slab of the band
 *             (diagonal-major), its window of the previous posterior row and nothing else in the LDS, evaluates every cell
 *             inside the band and none outside, and receives the hl + hr rows it needs from its two neighbours as
 *             self-validating 16-byte granules while it works on the cells that need only its own rows (band_forward.hpp).
 * AUTO: SMALL up to 64 states (up to 256 for batches that are not huge); else RESIDENT when the call's tiles fill more than half the compute units; CLUSTER for any other batch of more than
 * 16 items (64 <= S <= 4096: one batch = one forward launch); HELD up to three items (eight above 2048 states); ROWS for
 * 6..16 items (and above 2048 states); DENSE for large batches outside 64..4096 states; else GENERIC.  (The Python layer
 * adds what it knows about the matrix: DENSE for one batch with a narrow band or with scans too deep to prune.)
 *
 * A call selects a path in its `flags` (TORBI_HIP_PATH_FLAG); calls without one use the process-wide
 * default (torbi_hip_set_forward_path, initially the environment variable
 * TORBI_HIP_FORWARD=dense|pruned|resident|cluster|held, else AUTO).  A path that does not cover the shape falls
 * back as AUTO would.  A workspace of torbi_hip_workspace_bytes() fits every path.
 * torbi_hip_forward_path_on reports what a (B, S) batch would run on `device` with `flags`:
 * 0 generic, 1 dense, 3 resident, 4 rows, 5 cluster, 6 held, 7 small (2 is retired; 8 = band is reported by
 * torbi_hip_viterbi_decode_banded's phase_ms[3] and by the route record only); torbi_hip_forward_path is the same for device 0,
 * flags 0.
 */
#define TORBI_HIP_FORWARD_AUTO 0
#define TORBI_HIP_FORWARD_DENSE 1
#define TORBI_HIP_FORWARD_PRUNED 2
#define TORBI_HIP_FORWARD_RESIDENT 3
#define TORBI_HIP_FORWARD_CLUSTER 4
#define TORBI_HIP_FORWARD_HELD 5       /* B <= 16, S <= 4096: ONE launch, the matrix held in registers across the chip */
#define TORBI_HIP_FORWARD_BAND 6       /* (ABI 14) banded matrices through torbi_hip_viterbi_decode_banded; elsewhere = AUTO */
#define TORBI_HIP_PATH_FLAG(path) (((unsigned)(path) + 1u) << 4)   /* bits 4..6 of `flags`; 0 = process default */
int torbi_hip_set_forward_path(int path);
int torbi_hip_forward_path(int B, int S);
int torbi_hip_forward_path_on(int B, int S, int device, unsigned flags);
/*
 * (ABI 11) Name of the forward kernel the CALLING THREAD's most recent decode launched, spelled the way rocprofv3
 * prints kernels ("streamed::streamed_forward_kernel<15, 6>", "resident::resident_forward_kernel<12, 1, true, 1, true, 16, false>",
 * "dense::step_dense_kernel<8, 6, 8, 12, 8>", ...); empty before the first decode.  Measurement plumbing: bench.py
 * reports counter-derived figures (profiles/ *_pmc.json) only when they were taken on the kernel that is running.
 * No counterpart in the reference.
 */
int torbi_hip_last_forward_kernel(char *name_out, size_t capacity);

/*
 * The operator.  Replaces viterbi_decode_cuda (viterbi.cu:309-362) = forward trellis
 * kernel (:48-130) + argmax/repeat fill (:347-350) + backtrace kernel (:150-176).
 *
 *   observation   (B,T,S) fp32, log space          batch_frames (B) int32
 *   transition    (S,S)   fp32, [next, prev]       initial      (S) fp32
 *   indices_out   (B,T)   int32  -- fully overwritten
 *   workspace     >= torbi_hip_workspace_bytes(B,T,S) bytes, 256-byte aligned
 */
int torbi_hip_viterbi_decode(const float *observation, const int32_t *batch_frames,
                             const float *transition, const float *initial,
                             int32_t *indices_out, void *workspace, size_t workspace_bytes,
                             int B, int T, int S, int device, void *stream);

/*
 * The operator with flags.  TORBI_HIP_REUSE_TRANSITION: the caller promises that the previous call
 * that used `workspace` had the same B, T, S, forward path and transition CONTENTS and that nothing
 * else has written to the workspace since; the per-transition preparation (sorted transition rows /
 * packed panels, 0.2 ms at S = 1440) is then taken from the workspace instead of being rebuilt.
 * A serving loop that decodes batch after batch with one matrix sets it from the second batch on
 * (torbi_amd.DecodePipeline does).  TORBI_HIP_PATH_FLAG(path) selects the forward path for THIS call
 * (nothing process-wide is read or written: calls from different host threads on different
 * streams/devices are independent).  flags = 0 is torbi_hip_viterbi_decode.  Unknown bits: EINVAL.
 */
#define TORBI_HIP_REUSE_TRANSITION 1u
#define TORBI_HIP_COLLECT_STATS 2u     /* accepted and ignored since round 4 (the time-resident forms always leave statistics) */
#define TORBI_HIP_SHORTEST_FIRST 256u  /* RESIDENT: workgroups in ascending order of their items' lengths (default: longest first) */
#define TORBI_HIP_FEW_SEEDS 512u       /* RESIDENT / CLUSTER: ONE explicit candidate per item (its largest posterior) instead of
                                        * three.  Same results.  For callers that have seen shallow scans with this matrix
                                        * (torbi_hip_scan_stats: ~11 of 90 list blocks per scan on the benchmark): the seed
                                        * gathers walk whole rows of the transposed matrix through the L2s -- two thirds of the
                                        * kernel's memory-side read traffic with three seeds -- and buy nothing on flat
                                        * posterior rows; on peaked rows three seeds scan a fifth fewer list blocks. */
#define TORBI_HIP_MANY_SEEDS 1024u     /* ... and THREE.  Without either flag the whole-tile form keeps three seeds and the
                                        * cluster form one (its passes run in lock step across the workgroup, so the seed
                                        * gathers' latency sits on every timestep's critical path: 512 x 1440 18.8 against
                                        * 20.7 us per timestep; peaked rows with a dense matrix are 10 % faster with three) */
int torbi_hip_viterbi_decode_ex(const float *observation, const int32_t *batch_frames,
                                const float *transition, const float *initial,
                                int32_t *indices_out, void *workspace, size_t workspace_bytes,
                                int B, int T, int S, int device, void *stream, unsigned flags);

/*
 * Several batches that share `transition`/`initial`, decoded together.  Replaces a LOOP over the
 * operator (reference torbi/core.py:417-457 decodes batch after batch): batch items are independent
 * (viterbi.cpp:65, viterbi.cu:58), so the batches of a many-file job can share one set of launches.
 * On the RESIDENT path (the AUTO choice when the group's items fill half the compute units) the whole
 * group is ONE forward launch and ONE backtrace launch; the per-transition preparation is built once,
 * in the first non-empty batch's workspace.  Otherwise the batches are decoded one after the other on
 * `stream`, each on the path it would take alone.  Every batch needs its own workspace of
 * torbi_hip_workspace_bytes(B, T, S).  Batches may differ in B and T; count <= TORBI_HIP_MAX_BATCHES.
 *
 * phase_ms: NULL, or a HOST pointer to 6 floats -- the call then brackets its phases with hipEvents on
 * `stream` and SYNCHRONISES it (bench.py):
 *   [0] forward recurrence incl. preparation, ms   [1] argmax + backtrace, ms
 *   [2] forward kernel launches                    [3] route that ran (0 generic .. 7 small)
 *   [4] per-transition preparation alone, ms       [5] batches the forward launch(es) covered
 * (for batches decoded one after the other [0],[1],[2],[4] describe the LAST batch).
 */
#define TORBI_HIP_MAX_BATCHES 16
typedef struct torbi_hip_batch {
    const float *observation;     /* (B,T,S) fp32 */
    const int32_t *batch_frames;  /* (B) int32 */
    int32_t *indices_out;         /* (B,T) int32, fully overwritten */
    void *workspace;              /* >= torbi_hip_workspace_bytes(B,T,S) bytes, 256-byte aligned */
    size_t workspace_bytes;
    int B, T;
} torbi_hip_batch;
int torbi_hip_viterbi_decode_batches(const torbi_hip_batch *batches, int count, const float *transition,
                                     const float *initial, int S, int device, void *stream, unsigned flags,
                                     float *phase_ms);

/*
 * (ABI 12) The same with the per-transition preparation of the time-resident routes (sorted and arranged transition rows,
 * transposed matrix: what TORBI_HIP_REUSE_TRANSITION is about) kept by the CALLER in `preparation`
 * (>= torbi_hip_preparation_bytes(S) bytes of device memory, 256-byte aligned; 25.6 MB at 1440 states) instead of inside the
 * first batch's workspace.  For callers that allocate scratch per call like the reference's operator does
 * (viterbi.cu:331-336): the workspace may be new every time, the preparation lives with the matrix.  With
 * TORBI_HIP_REUSE_TRANSITION the caller promises that `preparation` was filled by an earlier call with the same S and
 * transition CONTENTS on a route of the same tile size (S <= 2048 / above), ordered before this one; without the flag it
 * is (re)built.  Routes that keep no such preparation (dense, rows, generic, held) ignore the buffer AND the flag (they
 * prepare in the workspace as ever).  *filled (may be NULL) = 1 when `preparation` holds the matrix's preparation once the
 * enqueued work has run -- i.e. whether the NEXT call may pass the flag -- else 0.  preparation NULL = the plain call.
 */
size_t torbi_hip_preparation_bytes(int S);
int torbi_hip_viterbi_decode_batches_prepared(const torbi_hip_batch *batches, int count, const float *transition,
                                              const float *initial, int S, int device, void *stream, unsigned flags,
                                              float *phase_ms, void *preparation, size_t preparation_bytes, int *filled);

/*
 * (ABI 14) Banded transition matrices.  No counterpart in the reference's operator, which walks all S x S cells whatever
 * the matrix holds (viterbi.cpp:81-104, viterbi.cu:89-117); its evaluation workload is banded all the same
 * (torbi/evaluate/core.py:24-33: at most 175 finite entries per 1440-state row).
 *
 * torbi_hip_band_reach: the smallest reach_left / reach_right such that transition[j][i] == -inf whenever i < j - reach_left
 * or i > j + reach_right (-1 / -1 for a matrix without any finite entry: pass 0 / 0 on).  Runs one small kernel over the matrix on
 * `stream` and SYNCHRONISES it: call it once per matrix and keep the answer.
 *
 * torbi_hip_band_members: workgroups per 16-item tile the band kernel would use for `items` sequences with that band on
 * `device` (1 .. 16), or 0 when it does not cover the shape -- S % 4 == 0, 64 <= S <= 3072, a member's slab of the band
 * + window + merge buffer within the 160 KB LDS (at 1440 states: reach_left + reach_right <= 175 with 8 members per tile,
 * up to 16 members for fewer than 16 tiles), each reach at most a member's share of the states, reach_left + reach_right <= 508.
 *
 * (ABI 15) torbi_hip_band_members answers 1 where the band kernel runs WHOLE tiles: one workgroup per 16-item tile, the band
 * streamed from the L2 (S % 4 == 0, 64 <= S <= 1536), once the split form's launches of resident-sized pieces would add up to
 * more (8 * launches >= 5 * members; at 1440 states: from 129 tiles = 2 064 items), or for every other unit a tile where
 * the split form does not cover the band.
 *
 * torbi_hip_viterbi_decode_banded: torbi_hip_viterbi_decode_batches for a matrix whose band the caller states -- a
 * PROMISE, verified on the device (ABI 15): an entry outside the band that is not -inf is noticed by the launch that looks
 * at the matrix, and every item is then decoded again on the whole matrix (slow, identical to the reference).  Same arguments,
 * same workspaces (torbi_hip_workspace_bytes covers the route), same phase_ms (phase_ms[3] = 8 when the band kernel ran).
 * `transition` and the observations must be 16-byte aligned for the band kernel.  The band kernel runs for TORBI_HIP_FORWARD_AUTO and
 * TORBI_HIP_FORWARD_BAND when its plan covers the group -- under AUTO except for shapes SMALL decodes and for the handful
 * of sequences HELD takes --; otherwise, and for every other named path, the call IS torbi_hip_viterbi_decode_batches
 * (BAND named: as AUTO).  Waits inside the launch are bounded and repaired as in the CLUSTER form (give-ups are counted
 * in torbi_hip_scan_stats [127]; TORBI_HIP_CLUSTER_WAIT_US overrides the budget).
 */
int torbi_hip_band_reach(const float *transition, int S, int device, void *stream, int *reach_left_out, int *reach_right_out);
int torbi_hip_band_members(int items, int S, int reach_left, int reach_right, int device);
int torbi_hip_viterbi_decode_banded(const torbi_hip_batch *batches, int count, const float *transition, const float *initial,
                                    int S, int reach_left, int reach_right, int device, void *stream, unsigned flags,
                                    float *phase_ms);

/*
 * (ABI 15) A band with ONE CONSTANT outside it.  The reference's evaluation does not decode with log(p) but with
 * log(p + tiny) (torbi/evaluate/core.py:97-103 -> torbi/core.py:341-347): its pitch matrix holds log(tiny) = -87.34 outside the
 * band, not -inf, and candidates from out there do win on rows whose posteriors fall that far.  Every such candidate is
 * fl(post[i] + c), so the best of them is fl(M + c), M the largest posterior outside the band (rounding is monotone): the
 * whole-tile band kernel keeps every row's maximum and where it is attained, and decides every output exactly
 * (csrc/band_tile_forward.hpp; what it cannot decide -- in-band entries BELOW the constant where the row's maximum stands --
 * is decoded again in the reference's order).  The `_over` entry points are the banded ones with `background` = that constant
 * (-inf: identical to them): torbi_hip_band_reach_over takes the matrix's corner entry transition[0][S - 1] for it (bit for
 * bit) and answers the reach over every other value -- S - 1 / S - 1 ("no band") for a finite constant unless every other
 * entry lies ABOVE it, as in a pitch matrix: where the band reaches down to the constant, outputs next to a row's maximum
 * cannot be decided from the maximum alone and the batch would be decoded again every time; torbi_hip_band_members_over answers like torbi_hip_band_members (the split
 * form exchanges its members' row maxima: csrc/band_forward.hpp, band_forward_kernel<true>).
 */
int torbi_hip_band_reach_over(const float *transition, int S, int device, void *stream, int *reach_left_out, int *reach_right_out,
                              float *background_out);
int torbi_hip_band_members_over(int items, int S, int reach_left, int reach_right, float background, int device);
int torbi_hip_viterbi_decode_banded_over(const torbi_hip_batch *batches, int count, const float *transition, const float *initial,
                                         int S, int reach_left, int reach_right, float background, int device, void *stream,
                                         unsigned flags, float *phase_ms);

/*
 * Scan statistics for adaptive path selection (torbi_amd/viterbi.py uses them): copies 128 uint32 to `stats_out`
 * (DEVICE pointer) on `stream`.  Which of the two records below is copied is decided ON THE DEVICE by the route the last
 * decode with `workspace` actually took (every decode stamps it behind its scratch layout: a batch decoded inside a
 * launch group takes the group's route, whatever its own shape and flags would have chosen); `flags` is accepted for
 * compatibility and ignored.  Zeros for a decode on another route.
 *   RESIDENT (always collected; `workspace` = the FIRST batch's workspace of the launch group):
 *     stats_out[0]       list blocks walked by the sampled wave passes (every 16th timestep)
 *     stats_out[64]      wave passes counted
 *     stats_out[127]     (CLUSTER) workgroups that gave up waiting for their cluster; 0 on any sane run
 *   RESIDENT / CLUSTER / BAND / DENSE (ABI 14): stats_out[120], [121] = shader-clock ticks and 100 MHz wall-clock ticks of the
 *                        forward kernel's workgroup 0 (same unit both; DENSE: of the last timestep's launch) -- their ratio
 *                        x 100 MHz is the clock the kernel was delivered under its own load (measurement plumbing: bench.py
 *                        prices the vector ALU's ceiling at it)
 *   RESIDENT / CLUSTER / BAND, one batch of at most 1024 sequences: stats_out[122], [123] = path steps of the backtrace and
 *                        steps walked AGAIN where its speculative segments met (csrc/lazy_backtrace.hpp, chase_segment)
 *   BAND: stats_out[127] = members that gave up waiting (as CLUSTER)
 *   HELD: stats_out[127] = workgroups that ran out of polls (the launch could not be resident as a whole); the decode
 *                        was then redone by the repair kernel and its results are correct.  0 on any sane run.
 * sum(first half) / sum(second half) = list blocks per scan: about 11 of the S/16 = 90 on the 1440-state benchmark;
 * near S/16 nothing is being pruned and the dense path is faster.  TORBI_HIP_EUNSUPPORTED when the shape takes
 * neither path.
 */
int torbi_hip_scan_stats(const void *workspace, size_t workspace_bytes, int B, int T, int S,
                         unsigned *stats_out, int device, void *stream, unsigned flags);

/*
 * The operator for a UNIFORM transition matrix (every entry == log_transition), i.e. the
 * reference's default when from_probabilities() is called without a transition
 * (torbi/core.py:175-180 builds torch.full((S,S), log(1/S)) and runs the generic recurrence).
 * With identical rows the backpointer is the same for every next state, so the decode is O(S)
 * per timestep, needs no scratch and streams the observations once (HBM-bound).  Results are
 * bit-identical to torbi_hip_viterbi_decode on the materialised matrix.
 * Covers S % 4 == 0, S <= 4096, 16-byte aligned observation/initial; otherwise returns
 * TORBI_HIP_EUNSUPPORTED and the caller materialises the matrix.
 */
int torbi_hip_viterbi_decode_uniform(const float *observation, const int32_t *batch_frames,
                                     float log_transition, const float *initial,
                                     int32_t *indices_out, int B, int T, int S, int device,
                                     void *stream);
/*
 * (ABI 13) The same with the observations given as PROBABILITIES, the way from_probabilities() receives them by default
 * (log_probs=False): every element goes through the reference's torch.log and the epsilon round trip log(exp(x) + tiny)
 * (torbi/core.py:189-197; bit-identical to torbi_hip_log_epsilon_clamp, i.e. to the torch ops) as it is read, so the whole
 * default call -- probabilities in, no transition given -- is ONE pass over the observations.  `probabilities` is not
 * written (upstream's torch.log is out of place too).  `initial` is in log space as before.
 */
int torbi_hip_viterbi_decode_uniform_probabilities(const float *probabilities, const int32_t *batch_frames,
                                                   float log_transition, const float *initial, int32_t *indices_out,
                                                   int B, int T, int S, int device, void *stream);

/*
 * Same operator, instrumented for bench.py: torbi_hip_viterbi_decode_batches for one batch with
 * `phase_ms` (a HOST pointer to 6 floats, see there); SYNCHRONISES the stream.
 */
int torbi_hip_viterbi_decode_profiled(const float *observation, const int32_t *batch_frames,
                                      const float *transition, const float *initial,
                                      int32_t *indices_out, void *workspace,
                                      size_t workspace_bytes, int B, int T, int S, int device,
                                      void *stream, unsigned flags, float *phase_ms);

/*
 * Test/diagnostic access to the final posterior rows the forward pass produced by the last
 * decode that used `workspace` (reference: the `posterior` tensor, viterbi.cu:334-336).
 * Copies (B,S) fp32 into `posterior_out` (device pointer) on `stream`.  Where the rows live depends on the route that
 * decode took; it is read on the device from the route record the decode left in the workspace (`flags` is accepted for
 * compatibility and ignored).
 */
int torbi_hip_read_posterior(const void *workspace, size_t workspace_bytes,
                             const int32_t *batch_frames, float *posterior_out,
                             int B, int T, int S, int device, void *stream, unsigned flags);

/*
 * In-place epsilon clamp of from_probabilities (reference torbi/core.py:193-197):
 *     x <- log(exp(x) + FLT_MIN)        three torch ops upstream (exp_, += tiny, log_)
 * fused into one pass over the (B,T,S) observation tensor.  Uses the same device math
 * functions as PyTorch-ROCm's elementwise kernels; tests/test_gpu_parity.py checks bit
 * equality with the three torch ops on the same device.
 */
int torbi_hip_epsilon_clamp(float *x, uint64_t count, int device, void *stream);

/*
 * The same for PROBABILITY inputs: out <- log(exp(log(p)) + FLT_MIN), i.e. upstream's torch.log(observation)
 * (torbi/core.py:189-191, out of place) and the clamp behind it in one pass instead of four.  `probabilities`
 * is not written unless `out` == `probabilities` (in place: the many-file driver's own copies of its files); both
 * pointers 16-byte aligned.
 */
int torbi_hip_log_epsilon_clamp(const float *probabilities, float *out, uint64_t count, int device, void *stream);

/*
 * Measurement helper (not part of the reference interface): fills dst[0..count) with the
 * deterministic synthetic scores of torbi_amd/synth.py -- value(k) = 0 - u24(hash(stream_id,
 * seed, start + k)) * 2^-20 -- so bench.py can build the 1.5 GB headline input in HBM
 * without a host round trip.  Bit-identical to the numpy definition (tested).
 */
int torbi_hip_fill_synthetic(float *dst, uint64_t count, uint64_t start, int stream_id,
                             int seed, int device, void *stream);

/*
 * (ABI 14) The host side of a many-file job -- reading the payloads of torch.save()d observations into a batch buffer,
 * writing the decoded sequences' files, opening a batch of files -- lives in libtorbi_cpu.so ALONE (include/torbi_cpu.h:
 * torbi_cpu_read_rows, torbi_cpu_write_files, torbi_cpu_open_heads).  Rounds 2-4 exported the same functions from this
 * library as torbi_hip_read_rows / _write_files / _open_heads; they touch no device, and their first call from a reader thread
 * used to wait behind the HIP runtime's start-up on the calling thread (0.35-0.65 s per early batch of a cold job).
 */

#ifdef __cplusplus
}
#endif
#endif /* TORBI_HIP_H */
