"""The N > 1 path on CPU: two gloo ranks, each renders its tile shard (the oracle stands in for the GPU
renderer, which is allowed inside tests/), packs its slab, ONE all_gather, unpack — the assembled frame
must be bit-identical to the single-rank frame.  Also checks the slab index math used by the kernels."""
import os
import subprocess
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_layout_is_a_partition(rt):
    from royaltracer_dx_amd import sharding
    for (w, h, ts, world) in [(200, 120, 32, 3), (1920, 1080, 64, 8), (64, 64, 64, 2), (70, 33, 16, 5)]:
        own = sharding.owner_map(w, h, ts, world)
        seen = np.zeros((h, w), np.int32)
        for r in range(world):
            x, y, ok = sharding.slot_pixels(w, h, ts, r, world)
            assert len(x) == sharding.layout(w, h, ts, world)["npl"]
            assert (own[y[ok], x[ok]] == r).all()
            seen[y[ok], x[ok]] += 1
        assert (seen == 1).all()
        img = np.random.default_rng(0).normal(size=(h, w, 4)).astype(np.float32)
        slabs = np.stack([sharding.pack(img, ts, r, world) for r in range(world)])
        assert np.array_equal(sharding.unpack(slabs, w, h, ts, world), img)
    assert sharding.layout(1920, 1080, 64, 8)["npl"] * 16 == 64 * 4096 * 16   # 510 tiles -> 64 per rank, 4.2 MB slab at 1080p


def test_block_grid_tie_goes_to_the_grid_with_fewer_columns(rt):
    """RTX_FLAG_BLOCK_TILES: of two rank grids with the same block perimeter the first factorisation in ascending gx wins — the loop of block_grid() in csrc/rtx_api.hip,
    mirrored by sharding.block_grid; the GPU tests compare the two sides tile by tile (test_restir_on_shards_*, blocks = 1), this pins the choice itself"""
    from royaltracer_dx_amd import sharding
    assert sharding.block_grid(8, 8, 2) == (1, 2)            # 8 + 4 either way: fewer columns
    assert sharding.block_grid(8, 8, 4) == (2, 2)            # the square grid has the smaller perimeter (4 + 4 < 8 + 2)
    assert sharding.block_grid(60, 34, 8) == (4, 2)          # 1080p in 32-px tiles on 8 ranks: 15 + 17 (480 x 544 px blocks)
    assert sharding.block_grid(5, 3, 4) == (2, 2)
    own = sharding.owner_map(256, 256, 32, 2, True)
    assert (own[:128] == 0).all() and (own[128:] == 1).all()  # two ranks on a square image: the split runs across rows (gx = 1, gy = 2)


WORKER = r'''
import os, sys, time, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as graft
rt = graft.load_package(); orc = graft.load_oracle()
from royaltracer_dx_amd import sharding
dist, rank, world = sharding.init_process_group("gloo")
W, H, TS = 96, 80, 32
scene = rt.Scene.cornell()
p = rt.Params(width=W, height=H, spp=2, max_bounces=5, nee_samples=1, flags=1, tile_size=TS, shard_rank=rank, shard_count=world)
o = orc.Oracle().load(scene, W / H)
t0 = time.perf_counter()
acc, cnt = o.render(p)
slab = torch.from_numpy(sharding.pack(acc, TS, rank, world))
dist.barrier()
allslabs = sharding.gather_slabs(dist, slab)
frame = sharding.unpack(allslabs.numpy(), W, H, TS, world)
dt = sharding.max_over_ranks(dist, time.perf_counter() - t0)
rays = sharding.sum_over_ranks(dist, cnt)
if rank == 0:
    whole, wc = o.render(p.copy(shard_rank=0, shard_count=1))
    assert np.array_equal(frame.view(np.uint32), whole.view(np.uint32)), "assembled frame differs"
    assert tuple(int(v) for v in rays) == wc, (rays, wc)
    assert dt > 0
    print("GLOO_OK", world, int(rays.sum()))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gloo_frame_assembly(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", OMP_NUM_THREADS="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            pr.kill(); out, _ = pr.communicate()
        outs.append(out)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
    assert "GLOO_OK 2" in outs[0], outs[0]


def test_halo_plan_is_symmetric_covers_the_dilated_rectangle_and_matches_the_c_abi(rt):
    """SURVEY 8(f1) halo exchange of the ReSTIR history (include/rtx.h: rtx_restir_halo_plan): for every pair of ranks the region r sends to q is the region q receives from r
    (same rectangle, same byte count); what a rank receives plus its own rectangle covers its rectangle dilated by the halo exactly once; the numpy mirror in
    sharding.halo_plan and the C-ABI agree peer by peer; a rank of the 8-rank 1080p deal has at most 8 peers and sends < 8 MB where the all-gather moves 292 MB of history to every rank."""
    from royaltracer_dx_amd import sharding
    for (w, h, ts, world, halo) in [(1920, 1080, 32, 8, 32), (1920, 1080, 32, 4, 20), (160, 96, 32, 4, 32), (200, 120, 16, 6, 24), (3840, 2160, 64, 8, 40), (96, 80, 32, 2, 32), (70, 33, 16, 5, 30)]:
        plans = [sharding.halo_plan(w, h, ts, r, world, halo) for r in range(world)]
        L = sharding.layout(w, h, ts, world, True)
        for r in range(world):
            peers, sb, rb = plans[r]
            cp, csb, crb = rt.restir_halo_plan(rt.Params(width=w, height=h, tile_size=ts, shard_rank=r, shard_count=world, flags=rt.FLAG_BLOCK_TILES), halo)
            assert (csb, crb) == (sb, rb) and len(cp) == len(peers)
            for a, b in zip(cp, peers):
                assert (a.rank, (a.send_x0, a.send_y0, a.send_x1, a.send_y1), (a.recv_x0, a.recv_y0, a.recv_x1, a.recv_y1), a.send_offset, a.send_bytes, a.recv_offset, a.recv_bytes) == \
                       (b["rank"], b["send"], b["recv"], b["send_offset"], b["send_bytes"], b["recv_offset"], b["recv_bytes"])
            own = sharding.block_rect(L, w, h, r)
            cover = np.zeros((h, w), np.int32)
            cover[own[1]:own[3], own[0]:own[2]] += 1
            for p in peers:
                back = [q for q in plans[p["rank"]][0] if q["rank"] == r]
                assert len(back) == 1 and back[0]["send"] == p["recv"] and back[0]["recv"] == p["send"] and back[0]["send_bytes"] == p["recv_bytes"]
                cover[p["recv"][1]:p["recv"][3], p["recv"][0]:p["recv"][2]] += 1
            want = np.zeros((h, w), np.int32)
            if own[0] < own[2] and own[1] < own[3]:
                want[max(own[1] - halo, 0):min(h, own[3] + halo), max(own[0] - halo, 0):min(w, own[2] + halo)] = 1
            assert np.array_equal(cover, want), (w, h, world, r)
    peers, sb, rb = sharding.halo_plan(1920, 1080, 32, 1, 8, 32)
    assert len(peers) <= 8 and sb < 8e6 and sharding.layout(1920, 1080, 32, 8, True)["npl"] * 140 * 8 > 2.9e8
    with pytest.raises(rt.RtxError):
        rt.restir_halo_plan(rt.Params(width=64, height=64, tile_size=32, shard_rank=0, shard_count=2), 32)        # not the block deal


HALO_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as graft
rt = graft.load_package()
from royaltracer_dx_amd import sharding
dist, rank, world = sharding.init_process_group("gloo")
W, H, TS, HALO = 160, 96, 32, 32
truth = np.random.default_rng(5).integers(0, 256, (H, W, 140), dtype=np.uint8)          # the whole image's history records (what an unsharded run would hold)
L = sharding.layout(W, H, TS, world, True); own = sharding.block_rect(L, W, H, rank)
mine = np.zeros_like(truth); mine[own[1]:own[3], own[0]:own[2]] = truth[own[1]:own[3], own[0]:own[2]]      # a rank holds its own rectangle after a frame
peers, sb, rb = sharding.halo_plan(W, H, TS, rank, world, HALO)
send = torch.from_numpy(sharding.halo_pack(mine, peers)); recv = torch.zeros(rb, dtype=torch.uint8)
assert send.numel() == sb
sharding.exchange_halo(dist, peers, send, recv)
sharding.halo_unpack(mine, peers, recv.numpy())
y0, y1, x0, x1 = max(own[1] - HALO, 0), min(H, own[3] + HALO), max(own[0] - HALO, 0), min(W, own[2] + HALO)
assert np.array_equal(mine[y0:y1, x0:x1], truth[y0:y1, x0:x1]), "the history is not valid in the rectangle + halo"
out = mine.copy(); out[y0:y1, x0:x1] = 0
assert not out.any(), "records outside the rectangle + halo were written"
print("HALO_OK", rank, len(peers), sb)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_halo_exchange_between_gloo_ranks(tmp_path, world):
    """the point-to-point exchange itself (sharding.exchange_halo: one isend + irecv per peer in one batch) with `world` gloo ranks on the CPU: after pack -> exchange -> unpack
    every rank holds the true records in its rectangle dilated by the halo and nothing elsewhere"""
    script = tmp_path / "halo_worker.py"
    script.write_text(HALO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29551 + world), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            pr.kill(); out, _ = pr.communicate()
        outs.append(out)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
    assert all("HALO_OK" in o for o in outs), outs


def test_bench_self_launch_plumbing_without_a_gpu():
    """`python bench.py --gpus N` with no launcher must start its own ranks BEFORE importing torch / touching the GPU, relay rank 0 and
    exit with the worst child code.  Without a GPU the children fail (rtx has no CPU fallback): the parent must reap them, not hang, and
    fail too; a launcher / --gpus mismatch is refused up front."""
    chk = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules; print('NO_TORCH')" % ROOT],
                         capture_output=True, text=True, timeout=120)
    assert chk.returncode == 0 and "NO_TORCH" in chk.stdout, chk.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=120,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "does not match WORLD_SIZE" in bad.stderr
    import torch
    if torch.cuda.is_available():
        return        # on a GPU box the full flow is tests/test_gpu_parity.py::test_bench_two_ranks_assemble_the_single_rank_frame
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--dist-backend", "gloo", "--device", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode != 0                     # children could not open a GPU
    assert "SystemExit: launch with" not in run.stderr
