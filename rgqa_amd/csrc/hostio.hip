// Host side of the input path: copies a batch's rows out of the mmap'ed feature store into a pinned staging
// buffer with several threads.  One row = one image's [36, 2048] features (147 KB f16 / 295 KB f32): numpy's single-threaded
// take() moves ~5 GB/s, below what one GPU consumes (19 k QA pairs/s x 295 KB = 5.6 GB/s from the f32 store).
// Host code only: no device kernels in this translation unit.
#include <stdint.h>
#include <string.h>
#include <thread>
#include <vector>
#include "common.h"
#include "../../include/rgqa.h"

extern "C" int rgqa_host_gather_rows(const void* src, size_t row_bytes, size_t n_src_rows, const int64_t* rows, int n, void* dst, int threads) {
    RGQA_REQUIRE(src != nullptr && rows != nullptr && dst != nullptr && n >= 0 && row_bytes > 0, "host_gather_rows: bad arguments");
    for (int i = 0; i < n; ++i)
        RGQA_REQUIRE(rows[i] >= 0 && (size_t)rows[i] < n_src_rows, "host_gather_rows: row %lld outside the store (%zu rows)", (long long)rows[i], n_src_rows);
    if (threads < 1) threads = 1;
    if (threads > n) threads = n > 0 ? n : 1;
    auto work = [&](int t) {
        for (int i = t; i < n; i += threads)
            memcpy(static_cast<char*>(dst) + (size_t)i * row_bytes, static_cast<const char*>(src) + (size_t)rows[i] * row_bytes, row_bytes);
    };
    if (threads == 1) { work(0); return RGQA_OK; }
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    return RGQA_OK;
}
