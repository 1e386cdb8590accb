#!/bin/bash
# The per-rank shard tables of DESIGN §5 / BASELINE.md (every shard r of N rendered alone on the one GPU): run through gpurun, ~3 minutes.
# usage: tools/shard_table.sh <tag>   -> gpurun_out/<tag>_shard_time.md   (copied to profiles/ by tools/collect_profiles.py)
tag=$1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
out=gpurun_out/${tag}_shard_time.md
{
  echo "# Per-rank frame times of sharded frames, each shard rendered alone on ONE MI355X (tools/shard_time.py; PREDICTIONS of what the slowest rank of an N-GPU frame costs, no gather)"
  echo
  python3 tools/shard_time.py cornell pt 1 2 4 8 tile=32 && python3 tools/shard_time.py cornell pt 8 tile=64 &&
  python3 tools/shard_time.py sponza pt 1 2 4 8 tile=32 && python3 tools/shard_time.py bistro pt 1 8 tile=32 &&
  python3 tools/shard_time.py sponza restir 1 2 4 8 blocks=1 tile=32 halo=32 && python3 tools/shard_time.py sponza restir 8 tile=64 &&
  python3 tools/shard_time.py sponza4k pt 8 tile=64 frames=2 &&
  python3 tools/shard_time.py cornell pt 1 2 4 8 native=1 && python3 tools/shard_time.py sponza pt 1 2 4 8 native=1
} > $out 2> gpurun_out/${tag}_shard_time.err
echo "shard table rc=$?"
