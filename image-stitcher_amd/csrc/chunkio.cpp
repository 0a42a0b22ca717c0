// sq_write_files: the chunk files of an OME-Zarr store, written by native threads.
//
// The store the reference writes (save_region_ome_zarr, stitcher.py:771-859: zarr v2, chunks (1,1,1,512,512)) is one file per
// chunk: a config-3 region's six levels are 176 000 files of about half a MB.  After the encoder moved to the device
// (csrc/blosc.hip) the wall of a files -> store run was this: 17 596 files of a 4-plane sample took 2 s of a 2.2 s run when
// every one of them went through Python's open / write / close under the interpreter lock (profiles/r04_e2e_split_probe.log),
// while the decode threads, the PCIe copies, the fusion and the encoder together need 0.4 s.  Here the caller hands over ONE
// buffer of packed frames with their offsets and ONE blob of NUL-terminated paths; n_threads workers take files from a shared
// counter and do open / write / close (directories are the caller's: a few hundred per region).
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

extern "C" int sq_write_files(const char *paths, const int64_t *path_offsets, const void *data, const int64_t *data_offsets,
                              int64_t n_files, int32_t n_threads, int64_t *bytes_written) {
    if (n_files < 0 || (n_files && (!paths || !path_offsets || !data_offsets))) return sq::fail(SQ_ERR_INVALID, "sq_write_files: NULL argument");
    if (bytes_written) *bytes_written = 0;
    if (n_files == 0) return SQ_OK;
    for (int64_t i = 0; i < n_files; ++i)
        if (data_offsets[i + 1] < data_offsets[i]) return sq::fail(SQ_ERR_INVALID, "sq_write_files: data offsets of file %lld decrease", (long long)i);
    if (!data && data_offsets[n_files] > data_offsets[0]) return sq::fail(SQ_ERR_INVALID, "sq_write_files: NULL data");
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : 16, n_files));
    std::atomic<int64_t> next{0}, total{0};
    std::atomic<int> failed{0};
    std::string first_error;
    std::mutex error_mutex;
    auto work = [&]() {
        int64_t mine = 0;
        for (;;) {
            const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_files || failed.load(std::memory_order_relaxed)) break;
            const char *path = paths + path_offsets[i];
            const char *p = static_cast<const char *>(data) + data_offsets[i];
            int64_t left = data_offsets[i + 1] - data_offsets[i];
            const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
            int err = fd < 0 ? errno : 0;
            while (!err && left > 0) {
                const ssize_t w = ::write(fd, p, (size_t)left);
                if (w < 0) {
                    if (errno == EINTR) continue;
                    err = errno;
                    break;
                }
                p += w;
                left -= w;
                mine += w;
            }
            if (fd >= 0 && ::close(fd) != 0 && !err) err = errno;
            if (err) {
                std::lock_guard<std::mutex> lock(error_mutex);
                if (!failed.exchange(1)) first_error = std::string(path) + ": " + std::strerror(err);
                break;
            }
        }
        total.fetch_add(mine, std::memory_order_relaxed);
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve(nt - 1);
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    } catch (...) {      // fewer threads than asked for: the ones there are (and this one) take all the files
    }
    work();
    for (auto &th : pool) th.join();
    if (bytes_written) *bytes_written = total.load();
    if (failed.load()) return sq::fail(SQ_ERR_INVALID, "sq_write_files: %s", first_error.c_str());
    return SQ_OK;
}
