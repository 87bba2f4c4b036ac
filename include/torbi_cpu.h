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
 * operator, NaN and +inf inputs included: an item that produced a NaN / +inf posterior value -- every item when the
 * matrix or the initial vector hold one -- is decoded again in the reference's own order of evaluation,
 * viterbi.cpp:94-100, 218), thread count passed explicitly.
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
 * Host side of a many-file job, for GPU jobs and gpu=None alike (no device is touched; all pointers are HOST pointers;
 * plain pread / write on native threads, so libtorbi_cpu.so keeps linking nothing but libgomp / libstdc++ and a reader thread's
 * first call never meets the HIP runtime's start-up).
 *
 * torbi_cpu_read_rows: item k's `bytes[k]` bytes at `offsets[k]` of the open file `fds[k]` are read into `rows[k]`, and the
 * `zero_bytes[k]` bytes behind them are cleared, by `threads` native threads.  Replaces torch.load + pad_sequence per batch
 * (reference torbi/data/dataset.py:18-20, torbi/data/collate.py:24-31): the float32 payload of a torch.save()d observation
 * goes from the page cache to its row of the (pinned) batch buffer in one pass, outside the Python interpreter
 * (torbi_amd/fastio.py finds the payloads and builds the same batch tuples as the reference's collate).
 *
 * torbi_cpu_write_files: `count` whole files, file k = the `bytes[k]` bytes at `data[k]`, created or truncated at `paths[k]`.
 * Replaces the one-by-one torch.save of the reference's driver (torbi/core.py:449-457, 466-473: ~0.1 ms of interpreter time
 * per decoded sequence); the caller hands over finished torch.save containers (torbi_amd/fastio.py builds them from a
 * prebuilt image per length).
 *
 * torbi_cpu_open_heads: the step before read_rows: open `count` files and read the first `head_bytes` bytes of each (where a
 * torch.save container keeps its record headers and data.pkl; 4096 is enough): fds_out[k] (-1 where open() failed; the caller
 * closes every descriptor >= 0, also after an error), heads_out[k * head_bytes ..], lengths_out[k] = bytes actually read (a
 * short file reads short).  512 open + pread pairs take 30 ms under Python's interpreter lock and well under a millisecond here.
 *
 * All three return 0, TORBI_CPU_EINVAL, or TORBI_CPU_EIO_BASE - k = -(100 + k) for the first item k that could not be read /
 * written / opened in full, with its errno in *error_out (0 = the file ended early).
 */
#define TORBI_CPU_EIO_BASE (-100)
int torbi_cpu_read_rows(const int *fds, const int64_t *offsets, const int64_t *bytes, void *const *rows,
                        const int64_t *zero_bytes, int count, int threads, int *error_out);
int torbi_cpu_write_files(const char *const *paths, const void *const *data, const int64_t *bytes, int count,
                          int threads, int *error_out);
int torbi_cpu_open_heads(const char *const *paths, int count, int threads, int head_bytes, int *fds_out,
                         unsigned char *heads_out, int *lengths_out, int *error_out);

#ifdef __cplusplus
}
#endif
#endif /* TORBI_CPU_H */
