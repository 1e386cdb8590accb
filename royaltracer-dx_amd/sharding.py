"""Pixel-tile sharding across GPUs: host-side index math + the torch.distributed plumbing.

Tiles of `tile_size`^2 pixels are dealt round-robin to the ranks (tile t -> rank t % world; `blocks=True` = RTX_FLAG_BLOCK_TILES: one rectangle of tiles per rank); a rank's
compact "slab" holds its tiles in order, each tile as 8x8-pixel blocks (the order the kernels use for
their path slots).  The numpy functions here are the reference for k_pack_tiles / k_unpack_tiles and are
what the CPU (gloo) tests exercise; the GPU path calls rtx_pack_tiles / rtx_unpack_tiles instead.
No data-path collective exists inside a frame; the only exchange is ONE all_gather of the slabs at the end.
"""
import os
import numpy as np


def block_grid(tiles_x, tiles_y, world):
    """RTX_FLAG_BLOCK_TILES: the ranks as a gx x gy grid of tile rectangles, gx * gy = world with the smallest rectangle perimeter; ties go to the
    first factorisation in ascending gx, i.e. FEWER columns (block_grid() in csrc/rtx_api.hip is the same loop: the two must agree, or the ranks' slabs do not line up)"""
    best, g = None, (world, 1)
    for a in range(1, world + 1):
        if world % a:
            continue
        b = world // a
        cost = -(-tiles_x // a) + -(-tiles_y // b)
        if best is None or cost < best:
            best, g = cost, (a, b)
    return g


def layout(width, height, tile_size=64, world=1, blocks=False):
    ts = tile_size or 64
    if ts < 16 or ts > 1024 or ts & (ts - 1):          # the same rule as validate_tiling() behind rtx_render / rtx_pack_tiles / rtx_shard_slab_bytes
        raise ValueError("tile_size must be a power of two in [16, 1024] (0 = 64)")
    tiles_x, tiles_y = (width + ts - 1) // ts, (height + ts - 1) // ts
    total = tiles_x * tiles_y
    per = (total + world - 1) // world
    gx = gy = 0
    if blocks and world > 1:
        gx, gy = block_grid(tiles_x, tiles_y, world)
        per = -(-tiles_x // gx) * -(-tiles_y // gy)
    return dict(ts=ts, tiles_x=tiles_x, tiles_y=tiles_y, total=total, per=per, npl=per * ts * ts, gx=gx, gy=gy)


def shard_tiles(L, rank, world):
    """(tx, ty, valid) of the k-th tile of `rank`, k = 0 .. per - 1 — mirrors shard_tile in csrc/rtx_kernels.hpp"""
    k = np.arange(L["per"], dtype=np.int64)
    if L["gx"]:
        gx, gy, TX, TY = L["gx"], L["gy"], L["tiles_x"], L["tiles_y"]
        bx, by = rank % gx, rank // gx
        tx0, tx1, ty0, ty1 = bx * TX // gx, (bx + 1) * TX // gx, by * TY // gy, (by + 1) * TY // gy
        bw = -(-TX // gx)
        tx, ty = tx0 + k % bw, ty0 + k // bw
        return tx, ty, (tx < tx1) & (ty < ty1)
    t = rank + k * world
    return t % L["tiles_x"], t // L["tiles_x"], t < L["total"]


def slot_pixels(width, height, tile_size, rank, world, blocks=False):
    """(x, y, valid) for every local slot of `rank` — mirrors slot_to_pixel in csrc/rtx_dev_common.hpp"""
    L = layout(width, height, tile_size, world, blocks)
    ts = L["ts"]
    pl = np.arange(L["npl"], dtype=np.int64)
    k, r = pl // (ts * ts), pl % (ts * ts)
    ttx, tty, tok = shard_tiles(L, rank, world)
    tx, ty, ok = ttx[k], tty[k], tok[k].copy()
    bpr = ts // 8
    blk, ln = r // 64, r % 64
    x = tx * ts + (blk % bpr) * 8 + (ln % 8)
    y = ty * ts + (blk // bpr) * 8 + (ln // 8)
    ok &= (x < width) & (y < height)
    return x, y, ok


def pack(accum, tile_size, rank, world, blocks=False):
    h, w, _ = accum.shape
    x, y, ok = slot_pixels(w, h, tile_size, rank, world, blocks)
    slab = np.zeros((len(x), 4), np.float32)
    slab[ok] = accum[y[ok], x[ok]]
    return slab


def unpack(slabs, width, height, tile_size, world, out=None, blocks=False):
    """slabs: (world, npl, 4) -> (H, W, 4)"""
    if out is None:
        out = np.zeros((height, width, 4), np.float32)
    slabs = np.asarray(slabs).reshape(world, -1, 4)
    for r in range(world):
        x, y, ok = slot_pixels(width, height, tile_size, r, world, blocks)
        out[y[ok], x[ok]] = slabs[r][ok]
    return out


def owner_map(width, height, tile_size, world, blocks=False):
    ts = tile_size or 64
    tiles_x = (width + ts - 1) // ts
    yy, xx = np.mgrid[0:height, 0:width]
    if blocks and world > 1:
        own = np.full((height, width), -1, np.int64)
        for r in range(world):
            x, y, ok = slot_pixels(width, height, tile_size, r, world, True)
            own[y[ok], x[ok]] = r
        return own
    return ((yy // ts) * tiles_x + (xx // ts)) % world


# ---- ReSTIR on shards: halo exchange of the history (include/rtx.h: rtx_restir_halo_plan / pack_halo / unpack_halo) ----
def block_rect(L, width, height, rank):
    """pixel rectangle (x0, y0, x1, y1) of `rank` in the block deal — the same rule as shard_tiles, clipped to the image"""
    gx, gy, TX, TY, ts = L["gx"], L["gy"], L["tiles_x"], L["tiles_y"], L["ts"]
    bx, by = rank % gx, rank // gx
    return (min(width, (bx * TX // gx) * ts), min(height, (by * TY // gy) * ts), min(width, ((bx + 1) * TX // gx) * ts), min(height, ((by + 1) * TY // gy) * ts))


HALO_RECORD_BYTES = 140          # Reservoir_DI 40 | Reservoir_GI 40 | SampleData 60


def halo_plan(width, height, tile_size, rank, world, halo):
    """numpy-side mirror of rtx_restir_halo_plan (csrc/rtx_api.hip: halo_plan): [dict(rank, send=(x0, y0, x1, y1), recv=(...), send_offset, send_bytes, recv_offset,
    recv_bytes)] in ascending rank order, total send bytes, total receive bytes.  send = my rectangle ∩ the peer's dilated by `halo`; recv = the peer's ∩ mine dilated."""
    L = layout(width, height, tile_size, world, True)
    if world < 2 or not L["gx"]:
        raise ValueError("halo exchange needs world > 1 (block deal)")
    own = block_rect(L, width, height, rank)

    def clip(a, b):            # a ∩ dilate(b, halo)
        r = (max(a[0], max(b[0] - halo, 0)), max(a[1], max(b[1] - halo, 0)), min(a[2], min(width, b[2] + halo)), min(a[3], min(height, b[3] + halo)))
        return r if r[0] < r[2] and r[1] < r[3] else None
    peers, so, ro = [], 0, 0
    if own[0] >= own[2] or own[1] >= own[3]:
        return peers, 0, 0
    for q in range(world):
        if q == rank:
            continue
        rq = block_rect(L, width, height, q)
        if rq[0] >= rq[2] or rq[1] >= rq[3]:
            continue
        sr, rr = clip(own, rq), clip(rq, own)
        if sr is None or rr is None:
            continue
        sb, rb = (sr[2] - sr[0]) * (sr[3] - sr[1]) * HALO_RECORD_BYTES, (rr[2] - rr[0]) * (rr[3] - rr[1]) * HALO_RECORD_BYTES
        peers.append(dict(rank=q, send=sr, recv=rr, send_offset=so, send_bytes=sb, recv_offset=ro, recv_bytes=rb))
        so += sb; ro += rb
    return peers, so, ro


def halo_pack(history, peers):
    """history: (H, W, 140) uint8 records by pixel -> the rank's send buffer (what rtx_restir_pack_halo writes, records row-major per region)"""
    return np.concatenate([history[p["send"][1]:p["send"][3], p["send"][0]:p["send"][2]].reshape(-1) for p in peers]) if peers else np.zeros(0, np.uint8)


def halo_unpack(history, peers, recv):
    for p in peers:
        x0, y0, x1, y1 = p["recv"]
        history[y0:y1, x0:x1] = np.asarray(recv[p["recv_offset"]:p["recv_offset"] + p["recv_bytes"]]).reshape(y1 - y0, x1 - x0, HALO_RECORD_BYTES)
    return history


def exchange_halo(dist, peers, send, recv):
    """the frame's point-to-point exchange: one isend + one irecv per peer in ONE batch (RCCL groups them: every pair uses its own xGMI link); send / recv are 1-D uint8
    tensors on the device of the backend (cuda for nccl, cpu for gloo)"""
    ops = []
    for p in peers:
        ops.append(dist.P2POp(dist.isend, send[p["send_offset"]:p["send_offset"] + p["send_bytes"]], p["rank"]))
        ops.append(dist.P2POp(dist.irecv, recv[p["recv_offset"]:p["recv_offset"] + p["recv_bytes"]], p["rank"]))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return recv


# ---- torch.distributed plumbing (backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests) ----
def init_process_group(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist, rank, world


def gather_slabs(dist, slab, gathered=None):
    """the frame's single collective: every rank contributes its slab, every rank receives all of them"""
    import torch
    world = dist.get_world_size()
    if gathered is None:
        gathered = torch.empty((world * slab.numel(),), dtype=slab.dtype, device=slab.device)
    dist.all_gather_into_tensor(gathered, slab.reshape(-1))
    return gathered


def max_over_ranks(dist, seconds, device=None):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, values, device=None):
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
