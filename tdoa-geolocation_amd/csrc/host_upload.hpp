// host_upload.hpp -- host memory or file -> HBM at PCIe speed.
//
// Replaces the "one Read into a []byte" of loadIQData (processor.go:181-193).  A plain hipMemcpy
// from pageable memory reaches ~15 GB/s on the bench node and a single-threaded
// fread + DMA ~7 GB/s; here T worker threads each copy (memcpy or pread) their chunks into
// their own two pinned staging buffers and issue hipMemcpyAsync on their own stream, so the
// host copies run T wide and overlap the DMAs.  Buffers, streams and events are created once
// per context and reused.
#pragma once

#include <hip/hip_runtime.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace tdoa {

class StagedUploader {
public:
    static constexpr size_t kChunk = 8u << 20;

    ~StagedUploader() { destroy(); }

    int threads() const { return (int)streams_.size(); }

    // false: a HIP call failed (nothing is left half-initialised)
    bool init(int device, int n_threads)
    {
        if (!streams_.empty()) return true;
        device_ = device;
        n_threads = n_threads < 1 ? 1 : (n_threads > 16 ? 16 : n_threads);
        bool ok = true;
        for (int t = 0; t < n_threads && ok; t++) {
            hipStream_t s = nullptr;
            ok = hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
            if (ok) streams_.push_back(s);
            for (int k = 0; k < 2 && ok; k++) {
                void *p = nullptr;
                hipEvent_t e = nullptr;
                ok = hipHostMalloc(&p, kChunk, hipHostMallocDefault) == hipSuccess;
                if (ok) pinned_.push_back(p);
                ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
                if (e) events_.push_back(e);
            }
        }
        if (!ok) destroy();
        return ok;
    }

    void destroy()
    {
        for (void *p : pinned_) (void)hipHostFree(p);
        for (hipEvent_t e : events_) (void)hipEventDestroy(e);
        for (hipStream_t s : streams_) (void)hipStreamDestroy(s);
        pinned_.clear();
        events_.clear();
        streams_.clear();
    }

    // copy `bytes` from src (host memory) or, when src is NULL, from file descriptor fd at offset 0, to dst (device).
    // Returns 0 ok, 1 HIP failure, 2 short read.
    int run(uint8_t *dst, const uint8_t *src, int fd, size_t bytes)
    {
        const int T = threads();
        const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
        std::atomic<int> status{0};
        auto worker = [&](int t) {
            if (hipSetDevice(device_) != hipSuccess) { status = 1; return; }
            bool used[2] = {false, false};
            int k = 0;
            for (size_t c = (size_t)t; c < n_chunks && status == 0; c += (size_t)T, k ^= 1) {
                const size_t off = c * kChunk, len = bytes - off < kChunk ? bytes - off : kChunk;
                void *buf = pinned_[2 * t + k];
                hipEvent_t ev = events_[2 * t + k];
                if (used[k] && hipEventSynchronize(ev) != hipSuccess) { status = 1; return; }
                if (src) {
                    std::memcpy(buf, src + off, len);
                } else {
                    size_t got = 0;
                    while (got < len) {
                        const ssize_t r = pread(fd, static_cast<uint8_t *>(buf) + got, len - got, (off_t)(off + got));
                        if (r <= 0) { status = 2; return; }
                        got += (size_t)r;
                    }
                }
                if (hipMemcpyAsync(dst + off, buf, len, hipMemcpyHostToDevice, streams_[t]) != hipSuccess ||
                    hipEventRecord(ev, streams_[t]) != hipSuccess) { status = 1; return; }
                used[k] = true;
            }
            if (hipStreamSynchronize(streams_[t]) != hipSuccess) status = 1;
        };
        std::vector<std::thread> pool;
        const int active = (int)std::min<size_t>((size_t)T, n_chunks ? n_chunks : 1);
        for (int t = 1; t < active; t++) pool.emplace_back(worker, t);
        worker(0);
        for (auto &th : pool) th.join();
        if (status != 0)   // drain whatever the other streams still have in flight
            for (hipStream_t s : streams_) (void)hipStreamSynchronize(s);
        return status;
    }

private:
    int device_ = 0;
    std::vector<hipStream_t> streams_;
    std::vector<void *> pinned_;
    std::vector<hipEvent_t> events_;
};

}  // namespace tdoa
