"""Pixel-tile sharding across GPUs: host-side index math + the torch.distributed plumbing.

Tiles of `tile_size`^2 pixels are dealt round-robin to the ranks (tile t -> rank t % world); a rank's
compact "slab" holds its tiles in order, each tile as 8x8-pixel blocks (the order the kernels use for
their path slots).  The numpy functions here are the reference for k_pack_tiles / k_unpack_tiles and are
what the CPU (gloo) tests exercise; the GPU path calls rtx_pack_tiles / rtx_unpack_tiles instead.
No data-path collective exists inside a frame; the only exchange is ONE all_gather of the slabs at the end.
"""
import os
import numpy as np


def layout(width, height, tile_size=64, world=1):
    ts = tile_size or 64
    if ts < 16 or ts > 1024 or ts & (ts - 1):          # the same rule as validate_tiling() behind rtx_render / rtx_pack_tiles / rtx_shard_slab_bytes
        raise ValueError("tile_size must be a power of two in [16, 1024] (0 = 64)")
    tiles_x, tiles_y = (width + ts - 1) // ts, (height + ts - 1) // ts
    total = tiles_x * tiles_y
    per = (total + world - 1) // world
    return dict(ts=ts, tiles_x=tiles_x, tiles_y=tiles_y, total=total, per=per, npl=per * ts * ts)


def slot_pixels(width, height, tile_size, rank, world):
    """(x, y, valid) for every local slot of `rank` — mirrors slot_to_pixel in csrc/rtx_kernels.hip"""
    L = layout(width, height, tile_size, world)
    ts = L["ts"]
    pl = np.arange(L["npl"], dtype=np.int64)
    k, r = pl // (ts * ts), pl % (ts * ts)
    t = rank + k * world
    ok = t < L["total"]
    tx, ty = t % L["tiles_x"], t // L["tiles_x"]
    bpr = ts // 8
    blk, ln = r // 64, r % 64
    x = tx * ts + (blk % bpr) * 8 + (ln % 8)
    y = ty * ts + (blk // bpr) * 8 + (ln // 8)
    ok &= (x < width) & (y < height)
    return x, y, ok


def pack(accum, tile_size, rank, world):
    h, w, _ = accum.shape
    x, y, ok = slot_pixels(w, h, tile_size, rank, world)
    slab = np.zeros((len(x), 4), np.float32)
    slab[ok] = accum[y[ok], x[ok]]
    return slab


def unpack(slabs, width, height, tile_size, world, out=None):
    """slabs: (world, npl, 4) -> (H, W, 4)"""
    if out is None:
        out = np.zeros((height, width, 4), np.float32)
    slabs = np.asarray(slabs).reshape(world, -1, 4)
    for r in range(world):
        x, y, ok = slot_pixels(width, height, tile_size, r, world)
        out[y[ok], x[ok]] = slabs[r][ok]
    return out


def owner_map(width, height, tile_size, world):
    ts = tile_size or 64
    tiles_x = (width + ts - 1) // ts
    yy, xx = np.mgrid[0:height, 0:width]
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
