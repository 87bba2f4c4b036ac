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

#define TORBI_CPU_ABI_VERSION 3
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

/*
 * Host side of the many-file job for callers WITHOUT a HIP runtime (gpu=None): the same two entry points as
 * torbi_hip_read_rows / torbi_hip_write_files (include/torbi_hip.h; same arguments, same return codes:
 * 0, TORBI_CPU_EINVAL, or -(100 + index) of the first item that could not be read / written in full with its errno
 * in *error_out).  They replace torch.load + pad_sequence (torbi/data/dataset.py:18-20, collate.py:24-31) and the
 * per-file torch.save (torbi/core.py:466-473) of the reference's loop; plain pread / write on native threads, so
 * libtorbi_cpu.so keeps linking nothing but libgomp / libstdc++.
 */
#define TORBI_CPU_EIO_BASE (-100)
int torbi_cpu_read_rows(const int *fds, const int64_t *offsets, const int64_t *bytes, void *const *rows,
                        const int64_t *zero_bytes, int count, int threads, int *error_out);
int torbi_cpu_write_files(const char *const *paths, const void *const *data, const int64_t *bytes, int count,
                          int threads, int *error_out);
/* (torbi_hip_open_heads: the files' descriptors and first bytes, on native threads) */
int torbi_cpu_open_heads(const char *const *paths, int count, int threads, int head_bytes, int *fds_out,
                         unsigned char *heads_out, int *lengths_out, int *error_out);

#ifdef __cplusplus
}
#endif
#endif /* TORBI_CPU_H */
