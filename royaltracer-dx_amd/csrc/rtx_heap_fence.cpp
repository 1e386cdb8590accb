// rtx_heap_fence.cpp — TOOLING, empty in the product: `make VARIANT=fence VARFLAGS="-DRTX_DEBUG_FENCE -g -fno-omit-frame-pointer"` builds librtx_hip_fence.so, in which every
// C++ allocation of this library whose size lies in [RTX_FENCE_MIN, RTX_FENCE_MAX] bytes (environment, default 880 .. 944) lives on pages of its own: when it is deleted the
// pages are made inaccessible and never reused, so a write (or read) through a stale pointer faults AT THE INSTRUCTION that does it, and the handler below prints the native
// stack (resolve the librtx_hip_fence.so(+0x...) frames with llvm-symbolizer / addr2line on the same file).  Round 5 used it to hunt a stray write that changed two words of a
// 912-byte std::vector once in a few thousand contexts (tools/flaky_bisect.py; profiles/r05_determinism.md).
#ifdef RTX_DEBUG_FENCE
#include <execinfo.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <new>

namespace {
constexpr size_t kHdr = 64; constexpr uint64_t kMagic = 0x46454e4345525458ull;
struct Hdr { uint64_t magic; size_t bytes, map_bytes; };
size_t g_lo = 880, g_hi = 944; long g_budget = 400000; bool g_on = false;
void on_fault(int sig, siginfo_t* si, void*) {
    char line[160]; int n = snprintf(line, sizeof line, "\n[rtx fence] signal %d at address %p — native stack:\n", sig, si ? si->si_addr : nullptr);
    if (n > 0) (void)!write(2, line, (size_t)n);
    void* fr[64]; const int k = backtrace(fr, 64); backtrace_symbols_fd(fr, k, 2);
    signal(sig, SIG_DFL); raise(sig);
}
struct Init {
    Init() {
        if (const char* e = getenv("RTX_FENCE_MIN")) g_lo = (size_t)atol(e);
        if (const char* e = getenv("RTX_FENCE_MAX")) g_hi = (size_t)atol(e);
        if (const char* e = getenv("RTX_FENCE_PAGES")) g_budget = atol(e);
        struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = on_fault; sa.sa_flags = SA_SIGINFO | SA_NODEFER;
        sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr);
        g_on = true;
        fprintf(stderr, "[rtx fence] allocations of %zu .. %zu bytes are fenced (%ld pages at most)\n", g_lo, g_hi, g_budget);
    }
} g_init;
void* fenced_new(size_t n) {
    if (g_on && n >= g_lo && n <= g_hi && g_budget > 0) {
        const size_t map = ((kHdr + n + 4095) / 4096) * 4096;
        void* p = mmap(nullptr, map, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p != MAP_FAILED) { g_budget -= (long)(map / 4096); Hdr* h = (Hdr*)p; h->magic = kMagic; h->bytes = n; h->map_bytes = map; return (char*)p + kHdr; }
    }
    void* p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void fenced_delete(void* p) {
    if (!p) return;
    if (((uintptr_t)p & 4095u) == kHdr) { Hdr* h = (Hdr*)((char*)p - kHdr); if (h->magic == kMagic) { const size_t mb = h->map_bytes; madvise(h, mb, MADV_DONTNEED); mprotect(h, mb, PROT_NONE); return; } }      // never reused: a stale access faults
    free(p);
}
}  // namespace
#define RTX_LOCAL          // (the variant is linked with -Wl,-Bsymbolic-functions: the library binds its own calls to these, nobody else does)
RTX_LOCAL void* operator new(size_t n) { return fenced_new(n); }
RTX_LOCAL void* operator new[](size_t n) { return fenced_new(n); }
RTX_LOCAL void operator delete(void* p) noexcept { fenced_delete(p); }
RTX_LOCAL void operator delete[](void* p) noexcept { fenced_delete(p); }
RTX_LOCAL void operator delete(void* p, size_t) noexcept { fenced_delete(p); }
RTX_LOCAL void operator delete[](void* p, size_t) noexcept { fenced_delete(p); }
#endif
