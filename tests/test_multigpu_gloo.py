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
