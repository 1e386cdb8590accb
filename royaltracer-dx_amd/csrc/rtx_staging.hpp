// rtx_staging.hpp — every byte that moves between the CALLER's memory (pageable: a std::vector, a numpy array, a Go slice) and the device passes through two pinned chunks
// the context owns.  Round 5: once in a few thousand contexts a read-back into a freshly allocated pageable buffer (hipMemcpyAsync + hipStreamSynchronize) delivered a
// partly stale copy — a validator fed half of an older tree, a tree hash that differed while the tree rendered the right image (tools/flaky_validate.py, tools/d2h_probe.hip;
// profiles/r05_determinism.md).  Copies between pinned memory and the device have ONE well-defined meaning in every HIP runtime, pageable ones go through the runtime's
// pin-on-the-fly / staging heuristics; so:
//   to_device  memcpy into a pinned chunk, asynchronous copy from there.  The source is CONSUMED when the call returns (no lifetime rule for the caller to keep, no
//              synchronise), the copy itself is ordered on the stream like any other.
//   to_host    asynchronous copy into a pinned chunk, event wait, memcpy out — the next chunk's copy runs under the memcpy.  Complete when the call returns.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string.h>

namespace rtx {

struct Staging {
    static constexpr size_t kMinChunk = 64u << 10, kMaxChunk = 8u << 20;
    void* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false}; size_t chunk = 0; int next = 0;
    Staging() = default; Staging(const Staging&) = delete; Staging& operator=(const Staging&) = delete;
    ~Staging() { release(); }
    hipError_t wait(int k) { if (!busy[k]) return hipSuccess; busy[k] = false; return hipEventSynchronize(ev[k]); }
    hipError_t reserve(size_t bytes) {
        size_t want = kMinChunk; while (want < bytes && want < kMaxChunk) want <<= 1;
        if (want <= chunk) return hipSuccess;
        hipError_t e;
        for (int k = 0; k < 2; k++) { if ((e = wait(k)) != hipSuccess) return e; if (buf[k]) { (void)hipHostFree(buf[k]); buf[k] = nullptr; } }
        chunk = 0;
        for (int k = 0; k < 2; k++) {
            if ((e = hipHostMalloc(&buf[k], want, hipHostMallocDefault)) != hipSuccess) return e;
            if (!ev[k] && (e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)) != hipSuccess) return e;
        }
        chunk = want;
        return hipSuccess;
    }
    void release() {
        for (int k = 0; k < 2; k++) { (void)wait(k); if (buf[k]) (void)hipHostFree(buf[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); buf[k] = nullptr; ev[k] = nullptr; }
        chunk = 0;
    }
    hipError_t to_device(hipStream_t st, void* dst, const void* src, size_t bytes) {
        if (!bytes) return hipSuccess;
        hipError_t e = reserve(bytes); if (e != hipSuccess) return e;
        for (size_t off = 0; off < bytes; off += chunk) {
            const size_t n = bytes - off < chunk ? bytes - off : chunk; const int k = next; next ^= 1;
            if ((e = wait(k)) != hipSuccess) return e;                    // (the copy that last read this chunk)
            memcpy(buf[k], (const char*)src + off, n);
            if ((e = hipMemcpyAsync((char*)dst + off, buf[k], n, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
            if ((e = hipEventRecord(ev[k], st)) != hipSuccess) return e;
            busy[k] = true;
        }
        return hipSuccess;
    }
    hipError_t to_host(hipStream_t st, void* dst, const void* src, size_t bytes) {
        if (!bytes) return hipSuccess;
        hipError_t e = reserve(bytes); if (e != hipSuccess) return e;
        int pk = -1; size_t poff = 0, pn = 0;
        for (size_t off = 0; off < bytes; off += chunk) {
            const size_t n = bytes - off < chunk ? bytes - off : chunk; const int k = next; next ^= 1;
            if ((e = wait(k)) != hipSuccess) return e;                    // (an upload that still reads this chunk; a read-back's own chunk was emptied one round ago)
            if ((e = hipMemcpyAsync(buf[k], (const char*)src + off, n, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
            if ((e = hipEventRecord(ev[k], st)) != hipSuccess) return e;
            busy[k] = true;
            if (pk >= 0) { if ((e = wait(pk)) != hipSuccess) return e; memcpy((char*)dst + poff, buf[pk], pn); }
            pk = k; poff = off; pn = n;
        }
        if ((e = wait(pk)) != hipSuccess) return e;
        memcpy((char*)dst + poff, buf[pk], pn);
        return hipSuccess;
    }
};

}  // namespace rtx
