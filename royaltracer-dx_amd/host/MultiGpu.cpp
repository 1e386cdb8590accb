// MultiGpu.cpp — see MultiGpu.h.  Host code only (HIP runtime + RCCL); compiled into the rtx_render executable.
#include "MultiGpu.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <stdexcept>
#include <thread>

namespace {
void hipck(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
void ncclck(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }
}  // namespace

struct MultiGpuFrame::Impl {
    std::vector<rtx_ctx*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<void*> slab, gathered;          // per rank: its own slab, the slabs of all ranks
    std::vector<ncclComm_t> comm;
    size_t slab_bytes = 0;
};

MultiGpuFrame::MultiGpuFrame(const std::vector<int>& devices, Gather g) : m(new Impl), m_devices(devices), m_stats(devices.size()), m_gather(g) {
    if (devices.empty()) throw std::runtime_error("MultiGpuFrame: no devices");
    const int n = (int)devices.size();
    m->ctx.assign(n, nullptr); m->stream.assign(n, nullptr); m->slab.assign(n, nullptr); m->gathered.assign(n, nullptr);
    for (int r = 0; r < n; r++) {
        if (rtx_create(devices[r], &m->ctx[r]) != RTX_OK) throw std::runtime_error(std::string("rtx_create: ") + rtx_last_error(nullptr));
        hipck(hipSetDevice(devices[r]), "hipSetDevice");
        hipck(hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking), "hipStreamCreate");
        if (rtx_set_stream(m->ctx[r], m->stream[r]) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r]));   // pack -> gather -> unpack run stream-ordered
    }
    if (g == Gather::RCCL && n > 1) {
        m->comm.assign(n, nullptr);
        ncclck(ncclCommInitAll(m->comm.data(), n, devices.data()), "ncclCommInitAll");      // single process, one communicator per device
    }
}

MultiGpuFrame::~MultiGpuFrame() {
    for (size_t r = 0; r < m->ctx.size(); r++) {
        (void)hipSetDevice(m_devices[r]);
        if (m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
        if (r < m->comm.size() && m->comm[r]) (void)ncclCommDestroy(m->comm[r]);
        if (m->ctx[r]) rtx_destroy(m->ctx[r]);
        if (m->slab[r]) (void)hipFree(m->slab[r]);
        if (m->gathered[r]) (void)hipFree(m->gathered[r]);
        if (m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
    }
    delete m;
}

void MultiGpuFrame::SetScene(const Scene& s, float aspect) {
    std::vector<std::thread> th; std::vector<std::string> err(m->ctx.size());
    for (size_t r = 0; r < m->ctx.size(); r++)        // uploads + BVH builds run in parallel, one thread per context (SURVEY 8(b) threading contract)
        th.emplace_back([&, r] { if (UploadScene(s, m->ctx[r], aspect) != RTX_OK) err[r] = rtx_last_error(m->ctx[r]); });
    for (auto& t : th) t.join();
    for (auto& e : err) if (!e.empty()) throw std::runtime_error("MultiGpuFrame::SetScene: " + e);
}

void MultiGpuFrame::Render(const rtx_params& p0) {
    const int n = (int)m->ctx.size();
    rtx_params probe = p0; probe.shard_rank = 0; probe.shard_count = (uint32_t)n;
    size_t bytes = 0;
    if (rtx_shard_slab_bytes(&probe, &bytes) != RTX_OK) throw std::runtime_error(std::string("rtx_shard_slab_bytes: ") + rtx_last_error(nullptr));
    if (bytes != m->slab_bytes) {
        for (int r = 0; r < n; r++) {
            hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
            if (m->slab[r]) (void)hipFree(m->slab[r]);
            if (m->gathered[r]) (void)hipFree(m->gathered[r]);
            hipck(hipMalloc(&m->slab[r], bytes), "hipMalloc slab");
            hipck(hipMalloc(&m->gathered[r], bytes * (size_t)n), "hipMalloc gathered");
        }
        m->slab_bytes = bytes;
    }
    m_w = p0.width; m_h = p0.height;
    std::vector<std::string> err(n);
    const auto t0 = std::chrono::steady_clock::now();
    // phase 1, one thread per rank (rtx_render is synchronous; different contexts may be driven from different threads): render my tiles, enqueue the pack
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++) th.emplace_back([&, r] {
        try {
            hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
            rtx_params p = p0; p.shard_rank = (uint32_t)r; p.shard_count = (uint32_t)n;
            if (rtx_render(m->ctx[r], &p) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r]));
            (void)rtx_get_stats(m->ctx[r], &m_stats[r]);
            if (n > 1 && rtx_pack_tiles(m->ctx[r], &p, m->slab[r]) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r]));
        } catch (const std::exception& e) { err[r] = e.what(); }
    });
    for (auto& t : th) t.join();
    for (auto& e : err) if (!e.empty()) throw std::runtime_error("MultiGpuFrame::Render: " + e);
    if (n > 1) {
        // phase 2: ONE collective per frame, issued for all ranks of this process in one group, each on its rank's stream (behind that rank's pack)
        if (m_gather == Gather::RCCL) {
            ncclck(ncclGroupStart(), "ncclGroupStart");
            for (int r = 0; r < n; r++)
                ncclck(ncclAllGather(m->slab[r], m->gathered[r], bytes / sizeof(float), ncclFloat, m->comm[r], m->stream[r]), "ncclAllGather");
            ncclck(ncclGroupEnd(), "ncclGroupEnd");
        } else {                                                                        // testing stand-in (one GPU, several ranks): plain device copies
            for (int r = 0; r < n; r++) { hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); hipck(hipStreamSynchronize(m->stream[r]), "sync pack"); }
            for (int r = 0; r < n; r++) {
                hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
                for (int q = 0; q < n; q++)
                    hipck(hipMemcpyAsync((char*)m->gathered[r] + (size_t)q * bytes, m->slab[q], bytes, hipMemcpyDeviceToDevice, m->stream[r]), "copy slab");
            }
        }
        // phase 3: every rank scatters all slabs into its accumulation buffer (enqueued behind the gather on the same stream)
        for (int r = 0; r < n; r++) {
            rtx_params p = p0; p.shard_rank = (uint32_t)r; p.shard_count = (uint32_t)n;
            hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
            if (rtx_unpack_tiles(m->ctx[r], &p, m->gathered[r]) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r]));
        }
        for (int r = 0; r < n; r++) { hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); hipck(hipStreamSynchronize(m->stream[r]), "sync frame"); }
    }
    m_lastMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void MultiGpuFrame::Clear(uint32_t w, uint32_t h) {
    for (size_t r = 0; r < m->ctx.size(); r++) { hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); if (rtx_clear_accum(m->ctx[r], w, h) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r])); }
    m_w = w; m_h = h;
}
std::vector<float> MultiGpuFrame::ReadAccumulation(int rank) {
    std::vector<float> out((size_t)m_w * m_h * 4);
    hipck(hipSetDevice(m_devices[rank]), "hipSetDevice");
    if (rtx_read_accum(m->ctx[rank], out.data(), out.size() * 4) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[rank]));
    return out;
}
std::vector<uint8_t> MultiGpuFrame::ReadOutput(int rank) {
    std::vector<uint8_t> out((size_t)m_w * m_h * 4);
    hipck(hipSetDevice(m_devices[rank]), "hipSetDevice");
    if (rtx_read_srgb8(m->ctx[rank], out.data(), out.size()) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[rank]));
    return out;
}
