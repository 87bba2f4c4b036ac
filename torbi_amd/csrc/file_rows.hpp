// file_rows.hpp -- host side of the many-file job: byte ranges of files -> rows of a (pinned) batch buffer.
//
// The reference loads every observation file with torch.load and pads the batch with pad_sequence
// (torbi/data/dataset.py:18-20, torbi/data/collate.py:24-31) before anything moves to the device.  A decode of a
// 512-file batch takes a few milliseconds on an MI355X, so a many-file job runs at the speed of that host path.
// Here the float32 payload of each file is pread() from its place in the torch.save container straight into its
// row of the batch buffer by a handful of native threads (no Python, no GIL, one pass over the bytes); the rest of
// the row is zero-filled like collate's padding.  Plain host code: no device is touched.
#pragma once

#include <atomic>
#include <cerrno>
#include <cstring>
#include <stdint.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>

namespace filerows {

struct Job {
    const int *fds;
    const int64_t *offsets;      // where each payload starts in its file
    const int64_t *bytes;        // payload bytes to read
    void *const *rows;           // destination of each payload
    const int64_t *zero_bytes;   // bytes to clear behind each payload
    int n;
    std::atomic<int> next{0};
    std::atomic<int> failed{0};  // 1 + index of the first item that could not be read in full
    std::atomic<int> error{0};   // its errno (0 = the file ended early)
};

inline void work(Job *job) {
    for (;;) {
        const int k = job->next.fetch_add(1, std::memory_order_relaxed);
        if (k >= job->n) return;
        char *dst = static_cast<char *>(job->rows[k]);
        int64_t left = job->bytes[k], at = job->offsets[k];
        while (left > 0) {
            const ssize_t got = pread(job->fds[k], dst, (size_t)left, (off_t)at);
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) {
                int expected = 0;
                if (job->failed.compare_exchange_strong(expected, k + 1)) job->error.store(got < 0 ? errno : 0);
                break;
            }
            dst += got;
            at += got;
            left -= got;
        }
        if (job->zero_bytes[k] > 0) memset(static_cast<char *>(job->rows[k]) + job->bytes[k], 0, (size_t)job->zero_bytes[k]);
    }
}

// 0, or -(1 + index) of an item that could not be read in full (its errno in *error_out, 0 = short file)
inline int read_rows(const int *fds, const int64_t *offsets, const int64_t *bytes, void *const *rows,
                     const int64_t *zero_bytes, int n, int threads, int *error_out) {
    Job job;
    job.fds = fds; job.offsets = offsets; job.bytes = bytes; job.rows = rows; job.zero_bytes = zero_bytes; job.n = n;
    if (threads > n) threads = n;
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads - 1);
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, &job);
    work(&job);
    for (auto &t : pool) t.join();
    if (error_out) *error_out = job.error.load();
    const int failed = job.failed.load();
    return failed ? -failed : 0;
}

// ---- open a batch of files and read their first bytes (where a torch.save container keeps its record headers and
// data.pkl): 512 open() + pread() pairs cost 30 ms under Python's interpreter lock, a few hundred microseconds here.
struct HeadJob {
    const char *const *paths;
    int *fds;                    // out: descriptors (-1 where open failed)
    unsigned char *heads;        // out: head_bytes per file
    int *lengths;                // out: bytes read per file
    int head_bytes;
    int n;
    std::atomic<int> next{0};
    std::atomic<int> failed{0};
    std::atomic<int> error{0};
    std::atomic<long long> open_ns{0}, read_ns{0};      // (TORBI_FILE_TIMINGS: summed over the threads)
};

inline void head_work(HeadJob *job) {
    for (;;) {
        const int k = job->next.fetch_add(1, std::memory_order_relaxed);
        if (k >= job->n) return;
        job->lengths[k] = 0;
        const auto t_open = std::chrono::steady_clock::now();
        const int fd = open(job->paths[k], O_RDONLY | O_CLOEXEC);
        const auto t_read = std::chrono::steady_clock::now();
        job->open_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t_read - t_open).count();
        job->fds[k] = fd;
        int err = 0;
        if (fd < 0) {
            err = errno ? errno : EIO;
        } else {
            unsigned char *dst = job->heads + (size_t)k * job->head_bytes;
            int have = 0;
            while (have < job->head_bytes) {
                const ssize_t got = pread(fd, dst + have, (size_t)(job->head_bytes - have), (off_t)have);
                if (got < 0 && errno == EINTR) continue;
                if (got < 0) { err = errno ? errno : EIO; break; }
                if (got == 0) break;                 // a short file: the caller sees the length
                have += (int)got;
            }
            job->lengths[k] = have;
            job->read_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_read).count();
        }
        if (err) {
            int expected = 0;
            if (job->failed.compare_exchange_strong(expected, k + 1)) job->error.store(err);
        }
    }
}

// 0, or -(1 + index) of the first file that could not be opened / read (its errno in *error_out); descriptors that
// were opened stay open either way (the caller closes every fds_out[k] >= 0)
inline int open_heads(const char *const *paths, int n, int threads, int head_bytes, int *fds_out, unsigned char *heads_out,
                      int *lengths_out, int *error_out) {
    HeadJob job;
    job.paths = paths; job.fds = fds_out; job.heads = heads_out; job.lengths = lengths_out; job.head_bytes = head_bytes; job.n = n;
    if (threads > n) threads = n;
    if (threads < 1) threads = 1;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads - 1);
    for (int t = 1; t < threads; ++t) pool.emplace_back(head_work, &job);
    const auto t1 = std::chrono::steady_clock::now();
    head_work(&job);
    for (auto &t : pool) t.join();
    if (getenv("TORBI_FILE_TIMINGS")) {
        const auto ms = [](auto d) { return std::chrono::duration_cast<std::chrono::microseconds>(d).count() / 1000.0; };
        fprintf(stderr, "    open_heads: %d threads started in %.1f ms, all done after %.1f ms; open() %.1f ms, pread() %.1f ms summed\n",
                threads, ms(t1 - t0), ms(std::chrono::steady_clock::now() - t0), job.open_ns.load() / 1e6, job.read_ns.load() / 1e6);
    }
    if (error_out) *error_out = job.error.load();
    const int failed = job.failed.load();
    return failed ? -failed : 0;
}

// ---- the other direction: one small file per decoded sequence (torbi/core.py:449-457 saves them one by one) -------
struct WriteJob {
    const char *const *paths;
    const void *const *data;
    const int64_t *bytes;
    int n;
    std::atomic<int> next{0};
    std::atomic<int> failed{0};
    std::atomic<int> error{0};
};

inline void write_work(WriteJob *job) {
    for (;;) {
        const int k = job->next.fetch_add(1, std::memory_order_relaxed);
        if (k >= job->n) return;
        int err = 0;
        const int fd = open(job->paths[k], O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
        if (fd < 0) {
            err = errno ? errno : EIO;
        } else {
            const char *src = static_cast<const char *>(job->data[k]);
            int64_t left = job->bytes[k];
            while (left > 0) {
                const ssize_t put = write(fd, src, (size_t)left);
                if (put < 0 && errno == EINTR) continue;
                if (put <= 0) { err = errno ? errno : EIO; break; }
                src += put;
                left -= put;
            }
            if (close(fd) != 0 && !err) err = errno ? errno : EIO;
        }
        if (err) {
            int expected = 0;
            if (job->failed.compare_exchange_strong(expected, k + 1)) job->error.store(err);
        }
    }
}

// 0, or -(1 + index) of a file that could not be written (its errno in *error_out)
inline int write_files(const char *const *paths, const void *const *data, const int64_t *bytes, int n, int threads,
                       int *error_out) {
    WriteJob job;
    job.paths = paths; job.data = data; job.bytes = bytes; job.n = n;
    if (threads > n) threads = n;
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads - 1);
    for (int t = 1; t < threads; ++t) pool.emplace_back(write_work, &job);
    write_work(&job);
    for (auto &t : pool) t.join();
    if (error_out) *error_out = job.error.load();
    const int failed = job.failed.load();
    return failed ? -failed : 0;
}

}  // namespace filerows
