/*
 * torbi_cpu.h -- C ABI of libtorbi_cpu.so, the host twin of the MI355X decoder.
 *
 * The reference registers its operator for two device keys: CUDA (replaced by include/torbi_hip.h) and CPU,
 *
 *     viterbi_decode_cpu(observation, batch_frames, transition, initial)       torbi/csrc/viterbi.cpp:182-234
 *
 * which torbi.from_probabilities(..., gpu=None) selects (torbi/core.py:147-150) and torbi/viterbi.py:51-52 gives
 * its thread count through global torch state.  This entry point is that operator for callers who ask for the
 * CPU: same result contract as torbi_hip_viterbi_decode (decoded indices bit-identical to the reference CPU
 * operator for inputs without NaN), thread count passed explicitly.
 *
 * It is NOT a fallback: nothing in libtorbi_hip.so or in torbi_amd's GPU paths ever calls it; a GPU request without
 * a usable device raises.  It is also not the test oracle (oracle/ restates the reference's algorithm and cost
 * structure; this is an independent implementation -- value-only forward pass over item blocks with vectorised
 * (max,+) rows, backpointers recomputed along the decoded path -- checked against the same golden vectors).
 *
 *   - all pointers are HOST pointers; tensors contiguous row-major, fp32 / int32 as in torbi_hip.h
 *   - the call is synchronous; scratch ((B,T,S) fp32 posterior history, replacing the reference's int32 trellis of
 *     the same size) is allocated and freed inside
 *   - return value: 0 = success, TORBI_CPU_EINVAL, TORBI_CPU_ENOMEM
 */
#ifndef TORBI_CPU_H
#define TORBI_CPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TORBI_CPU_ABI_VERSION 1
#define TORBI_CPU_OK 0
#define TORBI_CPU_EINVAL (-1)   /* null pointer / non-positive dimension */
#define TORBI_CPU_ENOMEM (-6)   /* the posterior history could not be allocated */

int torbi_cpu_abi_version(void);

/*
 *   observation   (B,T,S) fp32, log space          batch_frames (B) int32 (clamped to [1, T])
 *   transition    (S,S)   fp32, [next, prev]       initial      (S) fp32
 *   indices_out   (B,T)   int32 -- fully overwritten
 *   num_threads   worker threads (OpenMP); <= 0 = the runtime's default
 */
int torbi_cpu_viterbi_decode(const float *observation, const int32_t *batch_frames, const float *transition,
                             const float *initial, int32_t *indices_out, int B, int T, int S, int num_threads);

#ifdef __cplusplus
}
#endif
#endif /* TORBI_CPU_H */
