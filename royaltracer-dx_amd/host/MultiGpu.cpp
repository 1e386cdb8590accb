// MultiGpu.cpp — see MultiGpu.h.  Host code only (HIP runtime + RCCL); compiled into the rtx_render executable.
#include "MultiGpu.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <mutex>
#include <utility>

// The ranks' streams are borrowed from a process-wide list and handed back idle, never destroyed — as the library does with its own (csrc/rtx_api.hip: StreamPool;
// profiles/r05_determinism.md: a process that creates and destroys HIP streams by the thousand gets, rarely, a stray write into its heap).
static std::mutex g_rank_stream_mu; static std::vector<std::pair<int, hipStream_t>> g_rank_streams;
static hipStream_t rank_stream_acquire(int device) {
    { std::lock_guard<std::mutex> g(g_rank_stream_mu); for (size_t i = 0; i < g_rank_streams.size(); i++) if (g_rank_streams[i].first == device) { hipStream_t st = g_rank_streams[i].second; g_rank_streams.erase(g_rank_streams.begin() + (long)i); return st; } }
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) throw std::runtime_error("MultiGpuFrame: hipStreamCreate failed");
    return st;
}
static void rank_stream_release(int device, hipStream_t st) { (void)hipStreamSynchronize(st); std::lock_guard<std::mutex> g(g_rank_stream_mu); g_rank_streams.emplace_back(device, st); }

namespace {
void hipck(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
void ncclck(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }
}  // namespace

struct MultiGpuFrame::Impl {
    std::vector<rtx_ctx*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<void*> slab, gathered;          // per rank: its own slab, the slabs of all ranks
    std::vector<void*> state_slab, state_gathered;      // ReSTIR frames: the history records (u3 / u5 / u7) of the rank's own tiles, 140 B per pixel
    std::vector<void*> halo_send, halo_recv; std::vector<size_t> halo_send_cap, halo_recv_cap;       // ... or, with SetHaloExchange, only the border strips (rtx_restir_pack_halo)
    std::vector<ncclComm_t> comm;
    size_t slab_bytes = 0, state_bytes = 0;
    // persistent per-rank worker threads (round 4: a frame used to create and join N std::threads): a worker sleeps on `cv` until `gen` moves, runs `job(rank)`, counts itself done
    std::vector<std::thread> worker;
    std::mutex mu; std::condition_variable cv, cv_done;
    const std::function<void(int)>* job = nullptr;
    unsigned long long gen = 0; int done = 0, nactive = 0; bool quit = false;
    std::vector<std::string> err;
};

MultiGpuFrame::MultiGpuFrame(const std::vector<int>& devices, Gather g, bool always_gather, int only_rank) : m(new Impl), m_devices(devices), m_stats(devices.size()), m_gather(g), m_always(always_gather), m_only(only_rank) {
    // a constructor that throws runs no destructor: whatever was created before the failure (contexts, streams, communicators, Impl itself) is torn down here
    try {
        if (devices.empty()) throw std::runtime_error("MultiGpuFrame: no devices");
        const int n = (int)devices.size();
        if (only_rank >= n || (only_rank >= 0 && g != Gather::COPY)) throw std::runtime_error("MultiGpuFrame: only_rank needs a rank of the list and Gather::COPY");
        m->ctx.assign(n, nullptr); m->stream.assign(n, nullptr); m->slab.assign(n, nullptr); m->gathered.assign(n, nullptr);
        m->state_slab.assign(n, nullptr); m->state_gathered.assign(n, nullptr);
        m->halo_send.assign(n, nullptr); m->halo_recv.assign(n, nullptr); m->halo_send_cap.assign(n, 0); m->halo_recv_cap.assign(n, 0);
        for (int r = 0; r < n; r++) {
            if (only_rank >= 0 && r != only_rank) continue;
            if (rtx_create(devices[r], &m->ctx[r]) != RTX_OK) throw std::runtime_error(std::string("rtx_create: ") + rtx_last_error(nullptr));
            hipck(hipSetDevice(devices[r]), "hipSetDevice");
            m->stream[r] = rank_stream_acquire(devices[r]);
            if (rtx_set_stream(m->ctx[r], m->stream[r]) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r]));   // render -> pack -> gather -> unpack run stream-ordered
            if (rtx_set_option(m->ctx[r], RTX_OPT_ASYNC, 1) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r]));   // rtx_render only enqueues: no host join before the pack
        }
        m->err.assign(n, std::string());
        for (int r = 0; r < n; r++) {
            if (!m->ctx[r]) continue;
            m->nactive++;
            m->worker.emplace_back([this, r] {
                unsigned long long seen = 0;
                for (;;) {
                    const std::function<void(int)>* job;
                    { std::unique_lock<std::mutex> lk(m->mu); m->cv.wait(lk, [&] { return m->quit || m->gen != seen; }); if (m->quit) return; seen = m->gen; job = m->job; }
                    try { (void)hipSetDevice(m_devices[r]); (*job)(r); } catch (const std::exception& e) { m->err[r] = e.what(); }
                    { std::lock_guard<std::mutex> lk(m->mu); m->done++; }
                    m->cv_done.notify_one();
                }
            });
        }
        if (g == Gather::RCCL && (n > 1 || always_gather)) {
            m->comm.assign(n, nullptr);
            ncclck(ncclCommInitAll(m->comm.data(), n, devices.data()), "ncclCommInitAll");      // single process, one communicator per device
        }
    } catch (...) { Teardown(); throw; }
}

void MultiGpuFrame::RunOnRanks(const std::function<void(int)>& job) {
    for (auto& e : m->err) e.clear();
    { std::lock_guard<std::mutex> lk(m->mu); m->job = &job; m->done = 0; m->gen++; }
    m->cv.notify_all();
    { std::unique_lock<std::mutex> lk(m->mu); m->cv_done.wait(lk, [&] { return m->done == m->nactive; }); m->job = nullptr; }
    for (auto& e : m->err) if (!e.empty()) throw std::runtime_error(e);
}

void MultiGpuFrame::Teardown() {
    if (!m) return;
    { std::lock_guard<std::mutex> lk(m->mu); m->quit = true; }
    m->cv.notify_all();
    for (auto& t : m->worker) if (t.joinable()) t.join();
    for (size_t r = 0; r < m->ctx.size(); r++) {
        if (!m->ctx[r] && !m->stream[r]) continue;
        (void)hipSetDevice(m_devices[r]);
        if (m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
        if (r < m->comm.size() && m->comm[r]) (void)ncclCommDestroy(m->comm[r]);
        if (m->ctx[r]) rtx_destroy(m->ctx[r]);
        for (std::vector<void*>* v : {&m->slab, &m->gathered, &m->state_slab, &m->state_gathered, &m->halo_send, &m->halo_recv}) if (r < v->size() && (*v)[r]) (void)hipFree((*v)[r]);
        if (m->stream[r]) rank_stream_release(m_devices[r], m->stream[r]);        // (never destroyed: see rank_stream_acquire)
    }
    delete m; m = nullptr;
}
MultiGpuFrame::~MultiGpuFrame() { Teardown(); }

// (re)allocate one slab + one gathered buffer per rank; on a failure part-way the sizes read 0 and every pointer is either valid or null
void MultiGpuFrame::EnsureSlabs(std::vector<void*>& slab, std::vector<void*>& gathered, size_t& have, size_t bytes) {
    if (bytes == have) return;
    const int n = (int)m->ctx.size();
    have = 0;
    for (int r = 0; r < n; r++) {
        if (!m->ctx[r]) continue;
        hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
        if (slab[r]) { (void)hipFree(slab[r]); slab[r] = nullptr; }
        if (gathered[r]) { (void)hipFree(gathered[r]); gathered[r] = nullptr; }
        hipck(hipMalloc(&slab[r], bytes), "hipMalloc slab");
        hipck(hipMalloc(&gathered[r], bytes * (size_t)n), "hipMalloc gathered");
    }
    have = bytes;
}

// ONE collective per call for all ranks of this process, each on its rank's stream (behind whatever that rank enqueued before: its pack kernel)
void MultiGpuFrame::AllGather(std::vector<void*>& slab, std::vector<void*>& gathered, size_t bytes) {
    const int n = (int)m->ctx.size();
    if (m_gather == Gather::RCCL) {
        ncclck(ncclGroupStart(), "ncclGroupStart");
        for (int r = 0; r < n; r++)
            ncclck(ncclAllGather(slab[r], gathered[r], bytes / sizeof(float), ncclFloat, m->comm[r], m->stream[r]), "ncclAllGather");
        ncclck(ncclGroupEnd(), "ncclGroupEnd");
    } else if (m_only >= 0) {                                                       // one rank of N measured alone: its own slab into its slot, stream-ordered (what the collective hands back to it)
        hipck(hipSetDevice(m_devices[m_only]), "hipSetDevice");
        hipck(hipMemcpyAsync((char*)gathered[m_only] + (size_t)m_only * bytes, slab[m_only], bytes, hipMemcpyDeviceToDevice, m->stream[m_only]), "copy slab");
    } else {                                                                        // testing stand-in (one GPU, several ranks): plain device copies
        for (int r = 0; r < n; r++) { hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); hipck(hipStreamSynchronize(m->stream[r]), "sync pack"); }
        for (int r = 0; r < n; r++) {
            hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
            for (int q = 0; q < n; q++)
                hipck(hipMemcpyAsync((char*)gathered[r] + (size_t)q * bytes, slab[q], bytes, hipMemcpyDeviceToDevice, m->stream[r]), "copy slab");
        }
    }
}

void MultiGpuFrame::SetScene(const Scene& s, float aspect) {
    // uploads + BVH builds run in parallel, one (persistent) thread per context (SURVEY 8(b) threading contract)
    RunOnRanks([&](int r) { if (UploadScene(s, m->ctx[r], aspect) != RTX_OK) throw std::runtime_error(std::string("MultiGpuFrame::SetScene: ") + rtx_last_error(m->ctx[r])); });
}

void MultiGpuFrame::Render(const rtx_params& p0) {
    const int n = (int)m->ctx.size();
    rtx_params probe = p0; probe.shard_rank = 0; probe.shard_count = (uint32_t)n;
    size_t bytes = 0;
    if (rtx_shard_slab_bytes(&probe, &bytes) != RTX_OK) throw std::runtime_error(std::string("rtx_shard_slab_bytes: ") + rtx_last_error(nullptr));
    EnsureSlabs(m->slab, m->gathered, m->slab_bytes, bytes);
    m_w = p0.width; m_h = p0.height;
    const auto t0 = std::chrono::steady_clock::now();
    // phase 1, on every rank's worker thread (different contexts may be driven from different threads): ENQUEUE my tiles' frame (RTX_OPT_ASYNC) and the pack behind it
    RunOnRanks([&](int r) {
        rtx_params p = p0; p.shard_rank = (uint32_t)r; p.shard_count = (uint32_t)n;
        if (rtx_render(m->ctx[r], &p) != RTX_OK) throw std::runtime_error(std::string("MultiGpuFrame::Render: ") + rtx_last_error(m->ctx[r]));
        if ((n > 1 || m_always) && rtx_pack_tiles(m->ctx[r], &p, m->slab[r]) != RTX_OK) throw std::runtime_error(std::string("MultiGpuFrame::Render: ") + rtx_last_error(m->ctx[r]));
    });
    if (n > 1 || m_always) {
        AllGather(m->slab, m->gathered, bytes);                                         // phase 2: ONE collective per frame
        // phase 3: every rank scatters all slabs into its accumulation buffer (enqueued behind the gather on the same stream)
        for (int r = 0; r < n; r++) {
            if (!m->ctx[r]) continue;
            rtx_params p = p0; p.shard_rank = (uint32_t)r; p.shard_count = (uint32_t)n;
            hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
            if (rtx_unpack_tiles(m->ctx[r], &p, m->gathered[r]) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r]));
        }
    }
    // the ONE host wait of the frame: every rank's stream (render -> pack -> gather -> unpack)
    for (int r = 0; r < n; r++) { if (!m->ctx[r]) continue; hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); hipck(hipStreamSynchronize(m->stream[r]), "sync frame"); }
    m_lastMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int r = 0; r < n; r++) if (m->ctx[r]) (void)rtx_get_stats(m->ctx[r], &m_stats[r]);          // (collects the counters of the finished frame)
}

void MultiGpuFrame::SetInstanceTransform(uint32_t instance, const float o2w[16]) {
    const auto t0 = std::chrono::steady_clock::now();
    RunOnRanks([&](int r) {
        if (rtx_set_instance_transform(m->ctx[r], instance, o2w) != RTX_OK || rtx_commit_scene(m->ctx[r]) != RTX_OK)
            throw std::runtime_error(std::string("MultiGpuFrame::SetInstanceTransform: ") + rtx_last_error(m->ctx[r]));
    });
    for (size_t r = 0; r < m->ctx.size(); r++) { if (!m->ctx[r]) continue; hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); hipck(hipStreamSynchronize(m->stream[r]), "sync refit"); }
    m_refitMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void MultiGpuFrame::RenderRestir(const rtx_params& p0) {
    const int n = (int)m->ctx.size();
    rtx_params probe = p0; probe.shard_rank = 0; probe.shard_count = (uint32_t)n;
    size_t bytes = 0, sbytes = 0;
    if (rtx_shard_slab_bytes(&probe, &bytes) != RTX_OK || rtx_restir_state_slab_bytes(&probe, &sbytes) != RTX_OK) throw std::runtime_error(std::string("slab bytes: ") + rtx_last_error(nullptr));
    const bool gather = n > 1 || m_always;
    const bool halo = gather && n > 1 && m_halo > 0;                    // border strips between neighbours instead of the all-gather of the history
    if (halo && !(p0.flags & RTX_FLAG_BLOCK_TILES)) throw std::runtime_error("MultiGpuFrame::RenderRestir: the halo exchange needs RTX_FLAG_BLOCK_TILES");
    std::vector<std::vector<rtx_halo_peer>> peers(n);
    std::vector<uint64_t> send_tot(n, 0), recv_tot(n, 0);
    if (halo) for (int r = 0; r < n; r++) {                             // the plan of EVERY rank (also of the ranks this process does not hold: only_rank)
        rtx_params p = p0; p.shard_rank = (uint32_t)r; p.shard_count = (uint32_t)n;
        uint32_t np = 0;
        if (rtx_restir_halo_plan(&p, m_halo, nullptr, 0, &np, &send_tot[r], &recv_tot[r]) != RTX_OK) throw std::runtime_error(std::string("halo plan: ") + rtx_last_error(nullptr));
        peers[r].resize(np);
        if (np && rtx_restir_halo_plan(&p, m_halo, peers[r].data(), np, &np, nullptr, nullptr) != RTX_OK) throw std::runtime_error(std::string("halo plan: ") + rtx_last_error(nullptr));
    }
    if (gather) { EnsureSlabs(m->slab, m->gathered, m->slab_bytes, bytes); if (!halo) EnsureSlabs(m->state_slab, m->state_gathered, m->state_bytes, sbytes); }
    if (halo) for (int r = 0; r < n; r++) {
        if (!m->ctx[r]) continue;
        hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
        auto grow = [&](void*& p, size_t& cap, size_t need) { if (need <= cap && p) return; if (p) { (void)hipFree(p); p = nullptr; cap = 0; } hipck(hipMalloc(&p, std::max<size_t>(need, 16)), "hipMalloc halo"); cap = need; };
        grow(m->halo_send[r], m->halo_send_cap[r], (size_t)send_tot[r]); grow(m->halo_recv[r], m->halo_recv_cap[r], (size_t)recv_tot[r]);
    }
    m_xbytes = 0;
    for (int r = 0; r < n; r++) m_xbytes = std::max<uint64_t>(m_xbytes, halo ? send_tot[r] : (gather ? sbytes : 0));
    m_w = p0.width; m_h = p0.height;
    const auto t0 = std::chrono::steady_clock::now();
    RunOnRanks([&](int r) {                                             // phase 1: the three passes on my tiles, then the packs enqueued behind them
        rtx_params p = p0; p.shard_rank = (uint32_t)r; p.shard_count = (uint32_t)n;
        if (rtx_render_restir(m->ctx[r], &p) != RTX_OK) throw std::runtime_error(std::string("MultiGpuFrame::RenderRestir: ") + rtx_last_error(m->ctx[r]));
        (void)rtx_get_stats(m->ctx[r], &m_stats[r]);
        if (!gather) return;
        const int rc = halo ? rtx_restir_pack_halo(m->ctx[r], &p, m_halo, m->halo_send[r]) : rtx_restir_pack_state(m->ctx[r], &p, m->state_slab[r]);
        if (rc != RTX_OK || rtx_pack_tiles(m->ctx[r], &p, m->slab[r]) != RTX_OK) throw std::runtime_error(std::string("MultiGpuFrame::RenderRestir: ") + rtx_last_error(m->ctx[r]));
    });
    if (gather) {
        if (m_gather == Gather::RCCL) {                                 // phase 2: the frame's ONE exchange: history (all-gather, or a send + receive per neighbour) + framebuffer tiles in one group
            ncclck(ncclGroupStart(), "ncclGroupStart");
            for (int r = 0; r < n; r++) {
                if (halo) for (const rtx_halo_peer& e : peers[r]) {
                    ncclck(ncclSend((const char*)m->halo_send[r] + e.send_offset, e.send_bytes, ncclChar, (int)e.rank, m->comm[r], m->stream[r]), "ncclSend halo");
                    ncclck(ncclRecv((char*)m->halo_recv[r] + e.recv_offset, e.recv_bytes, ncclChar, (int)e.rank, m->comm[r], m->stream[r]), "ncclRecv halo");
                }
                else ncclck(ncclAllGather(m->state_slab[r], m->state_gathered[r], sbytes / sizeof(float), ncclFloat, m->comm[r], m->stream[r]), "ncclAllGather state");
                ncclck(ncclAllGather(m->slab[r], m->gathered[r], bytes / sizeof(float), ncclFloat, m->comm[r], m->stream[r]), "ncclAllGather tiles");
            }
            ncclck(ncclGroupEnd(), "ncclGroupEnd");
        } else {
            if (!halo) AllGather(m->state_slab, m->state_gathered, sbytes);
            else if (m_only >= 0) {                                     // one rank of N measured alone: as many bytes as it would receive, copied from its own strips
                hipck(hipSetDevice(m_devices[m_only]), "hipSetDevice");
                const size_t nb = (size_t)std::min(send_tot[m_only], recv_tot[m_only]);
                if (nb) hipck(hipMemcpyAsync(m->halo_recv[m_only], m->halo_send[m_only], nb, hipMemcpyDeviceToDevice, m->stream[m_only]), "copy strips");
            } else {                                                    // testing stand-in (one GPU, several ranks): what the peer sends me == what I receive from it
                for (int r = 0; r < n; r++) { hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); hipck(hipStreamSynchronize(m->stream[r]), "sync pack"); }
                for (int r = 0; r < n; r++) {
                    hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
                    for (const rtx_halo_peer& e : peers[r]) {
                        const rtx_halo_peer* back = nullptr;
                        for (const rtx_halo_peer& q : peers[e.rank]) if (q.rank == (uint32_t)r) back = &q;
                        if (!back || back->send_bytes != e.recv_bytes) throw std::runtime_error("MultiGpuFrame::RenderRestir: halo plans of two ranks do not mirror each other");
                        hipck(hipMemcpyAsync((char*)m->halo_recv[r] + e.recv_offset, (const char*)m->halo_send[e.rank] + back->send_offset, e.recv_bytes, hipMemcpyDeviceToDevice, m->stream[r]), "copy strip");
                    }
                }
            }
            AllGather(m->slab, m->gathered, bytes);
        }
        for (int r = 0; r < n; r++) {                                   // phase 3: scatter both, stream-ordered behind the exchange
            if (!m->ctx[r]) continue;
            rtx_params p = p0; p.shard_rank = (uint32_t)r; p.shard_count = (uint32_t)n;
            hipck(hipSetDevice(m_devices[r]), "hipSetDevice");
            const int rc = halo ? rtx_restir_unpack_halo(m->ctx[r], &p, m_halo, m->halo_recv[r]) : rtx_restir_unpack_state(m->ctx[r], &p, m->state_gathered[r]);
            if (rc != RTX_OK || rtx_unpack_tiles(m->ctx[r], &p, m->gathered[r]) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r]));
        }
        for (int r = 0; r < n; r++) { if (!m->ctx[r]) continue; hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); hipck(hipStreamSynchronize(m->stream[r]), "sync frame"); }
    }
    m_lastMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
void MultiGpuFrame::SetCamera(const float view[16], const float proj[16]) {
    for (size_t r = 0; r < m->ctx.size(); r++) { if (!m->ctx[r]) continue; hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); if (rtx_set_camera(m->ctx[r], view, proj) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r])); }
}
void MultiGpuFrame::ResetRestir() {
    for (size_t r = 0; r < m->ctx.size(); r++) { if (!m->ctx[r]) continue; hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); if (rtx_restir_reset(m->ctx[r]) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r])); }
}
void MultiGpuFrame::SetOption(int option, int64_t value) {
    for (size_t r = 0; r < m->ctx.size(); r++) if (m->ctx[r] && rtx_set_option(m->ctx[r], option, value) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r]));
}

void MultiGpuFrame::Clear(uint32_t w, uint32_t h) {
    for (size_t r = 0; r < m->ctx.size(); r++) { if (!m->ctx[r]) continue; hipck(hipSetDevice(m_devices[r]), "hipSetDevice"); if (rtx_clear_accum(m->ctx[r], w, h) != RTX_OK) throw std::runtime_error(rtx_last_error(m->ctx[r])); }
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
