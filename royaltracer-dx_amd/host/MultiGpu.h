// MultiGpu.h — the native N-GPU frame of the C++ host layer (SURVEY 8(e), §5 last row): ONE process, one rtx_ctx per GPU, pixel tiles dealt round-robin
// (tile t -> rank t mod N; with RTX_FLAG_BLOCK_TILES in the params one rectangle of tiles per rank), every rank renders its tiles, packs them into a slab (rtx_pack_tiles) and the frame ends with ONE all-gather of the slabs
// over xGMI followed by rtx_unpack_tiles.  Nothing is exchanged inside the frame.  The reference is single-GPU (one device, one command queue:
// Renderer.cpp:106-254); where its Renderer::PopulateCommandList issues the frame (Renderer.cpp:646-673) a maintainer calls MultiGpuFrame::Render.
//
// The collective is RCCL in a single process: ncclCommInitAll over the device list, ncclAllGather on every rank's own stream inside
// ncclGroupStart / ncclGroupEnd (/opt/rocm/include/rccl/rccl.h:236,678,923).  RCCL is linked into the rtx_render executable only, never into
// librtx_hip.so.  For tests on ONE GPU (two ranks cannot share a device under RCCL) the gather can be replaced by plain device-to-device copies
// (Gather::COPY): the same slab layout, the same pack / unpack kernels, the same result.
#pragma once
#include <functional>
#include <string>
#include <vector>
#include "Scenes.h"

class MultiGpuFrame {
public:
    enum class Gather { RCCL, COPY };
    // always_gather: run pack -> collective -> unpack even with ONE rank (a one-rank communicator): exercises the RCCL path of the frame — communicator setup,
    // ncclGroupStart / ncclAllGather / ncclGroupEnd on the rank's stream, the slab round trip — on a machine with a single GPU
    // only_rank >= 0 (measurement on ONE GPU): the frame of an N-rank run as rank `only_rank` alone sees it — its worker thread, its enqueue-only render, its pack, the gather
    // (its own slab copied into its slot of the gathered buffer; Gather::COPY only), its unpack, one sync — so that per-rank times include the host path (tools/shard_time.py native=1)
    MultiGpuFrame(const std::vector<int>& devices, Gather g, bool always_gather = false, int only_rank = -1);
    ~MultiGpuFrame();
    void SetScene(const Scene& s, float aspect);          // replicated on every GPU (Bistro-class: 0.4 GB << 288 GB)
    // one frame: p.shard_rank / shard_count are filled per rank; the assembled accumulation buffer ends up on EVERY rank (all-gather)
    void Render(const rtx_params& p);
    // one frame of the reference's ReSTIR pipeline on N shards: every rank runs rtx_render_restir on its tiles (passes 1 + 2 on the tiles dilated by
    // the 20-px spatial radius, pass 3 on its own), then ONE group of collectives ends the frame: the all-gather of the history records of the own tiles (u3 / u5 / u7,
    // 140 B per pixel: the next frame's temporal pass reprojects to arbitrary pixels) and the all-gather of the framebuffer tiles.  p.spp must be 1.
    void RenderRestir(const rtx_params& p);
    // halo_px > 0 (needs RTX_FLAG_BLOCK_TILES in the params): the history is exchanged as BORDER STRIPS between neighbouring rectangles — rtx_restir_pack_halo, one ncclSend +
    // ncclRecv per peer (<= 8, each pair on its own xGMI link) in the frame's one group of collectives, rtx_restir_unpack_halo — instead of all-gathered: ~5-7 MB sent per rank
    // at 1080p on 8 ranks where the all-gather delivers 292 MB to every rank.  Valid while a frame's reprojection displacement stays below halo_px - 20 (32 = one tile: 12 px
    // of motion per frame); StaleHistoryReads() of the last frame says whether it did.  0 (default): all-gather.
    void SetHaloExchange(uint32_t halo_px) { m_halo = halo_px; }
    uint64_t StaleHistoryReads() const { uint64_t s = 0; for (const rtx_stats& st : m_stats) s += st.restir_stale_history_reads; return s; }   // of the last RenderRestir, all ranks: must be 0
    uint64_t LastExchangeBytes() const { return m_xbytes; }    // history bytes the busiest rank SENT in the last RenderRestir (all-gather: its slab; halo: its strips)
    // a moving instance (the reference re-sets instance 1 and refits its TLAS every frame: Renderer.cpp:444-452, 594): rtx_set_instance_transform + a transform-only
    // rtx_commit_scene on EVERY rank — the scene is replicated, so every rank refits its own copy of the tree on its GPU (k_refit_tris / k_refit_nodes)
    void SetInstanceTransform(uint32_t instance, const float o2w[16]);
    double LastRefitMs() const { return m_refitMs; }        // wall time of the last SetInstanceTransform: max over ranks
    void SetCamera(const float view[16], const float proj[16]);   // every rank (rtx_set_camera keeps the previous matrices for the reprojection)
    void ResetRestir();                                   // forget the ReSTIR history on every rank
    void SetOption(int option, int64_t value);            // rtx_set_option on every rank
    void Clear(uint32_t width, uint32_t height);          // zero every rank's accumulation buffer (view-change reset)
    std::vector<float> ReadAccumulation(int rank = 0);
    std::vector<uint8_t> ReadOutput(int rank = 0);
    rtx_stats Stats(int rank) const { return m_stats[rank]; }
    double LastFrameMs() const { return m_lastMs; }       // wall time of Render: max over ranks, gather included
    int Ranks() const { return (int)m_devices.size(); }
private:
    struct Impl;
    Impl* m;
    void Teardown();
    void RunOnRanks(const std::function<void(int)>& job);      // every active rank's persistent worker thread runs job(rank); returns when all have finished; rethrows the first error
    void EnsureSlabs(std::vector<void*>& slab, std::vector<void*>& gathered, size_t& have, size_t bytes);
    void AllGather(std::vector<void*>& slab, std::vector<void*>& gathered, size_t bytes);
    std::vector<int> m_devices;
    std::vector<rtx_stats> m_stats;
    Gather m_gather;
    bool m_always = false;
    double m_lastMs = 0.0, m_refitMs = 0.0;
    int m_only = -1;
    uint32_t m_w = 0, m_h = 0, m_halo = 0;
    uint64_t m_xbytes = 0;
};
