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
