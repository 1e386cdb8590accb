#!/bin/bash
# The builder / visiting-order experiments of round 4 as one table (host replay of the device traversal, no GPU): bash tools/bvh_lab_table.sh > profiles/r04_bvh_lab.md
cd "$(dirname "$0")/.."
make -s -C tools bvh_lab || exit 1
echo "# Work per ray on the host replay of the device traversal (tools/bvh_lab: the product's builder + csrc/rtx_scene_host.cpp replay_trace), round 4"
echo
echo "480 x 270 camera rays + four cosine-sampled bounces each, one NEE-like shadow ray per vertex; closest-hit rays in the device's octant order, WITHOUT the speculative schedule"
echo "(the GPU's own counters, bench.py extra.*.work_per_ray, read ~2 % more node steps).  cost = steps x 205 x 64 / 47 + tests x 70 x 64 / 24 VALU lane-slots (profiles/r02_traversal.md)."
echo
echo "| scene | builder / order | leaf entries | wide nodes | stack | wide SAH (node) | closest: steps / ray | tests / ray | cost | any-hit (slot order unless noted): steps / ray | tests / ray | cost | build s |"
echo "|---|---|---|---|---|---|---|---|---|---|---|---|---|"
row() { # scene label args...
  local sc=$1 label=$2; shift 2
  local out; out=$(tools/bvh_lab $sc "$@")
  local h; h=$(echo "$out" | grep "^$sc:"); local a; a=$(echo "$out" | grep "ALL closest")
  echo "| $sc | $label | $(echo $h | sed -E 's/.*refs ([0-9]+).*/\1/') | $(echo $h | sed -E 's/.*nodes8 ([0-9]+).*/\1/') | $(echo $h | sed -E 's/.*stack ([0-9]+).*/\1/') | $(echo $h | sed -E 's/.*SAH node ([0-9.]+).*/\1/') | $(echo $a | sed -E 's/.*closest: steps\/ray ([0-9.]+) tris\/ray ([0-9.]+) cost ([0-9]+).*/\1 | \2 | \3/') | $(echo $a | sed -E 's/.*shadow: steps\/ray ([0-9.]+) tris\/ray ([0-9.]+) cost ([0-9]+).*/\1 | \2 | \3/') | $(echo $h | sed -E 's/.*build ([0-9.]+)s.*/\1/') |"
}
for sc in sponza bistro garage; do
  row $sc "round-3 builder (16 bins, no re-insertion)" reinsert=0
  row $sc "**default**: + 2 re-insertion passes over the <= 200 000 largest nodes" 
  row $sc "full-sweep SAH everywhere, no re-insertion" reinsert=0 sweep=100000000
  row $sc "spatial splits alpha 1e-5, no re-insertion" reinsert=0 split=1e-5
  row $sc "spatial splits + re-insertion" split=1e-5
  row $sc "exact slot assignment, no re-insertion" reinsert=0 slot_assign=1
  row $sc "default, any-hit NEAREST octant first" any_order=1
  row $sc "default, any-hit FARTHEST octant first" any_order=2
done
echo
echo "## Where the node steps go, and what a better visiting order could save (default builder)"
echo
echo '```'
for sc in sponza bistro; do
  tools/bvh_lab $sc whatif=3 | grep -E "probe|depth|shadow rays:"
  tools/bvh_lab $sc lower=1 | grep -E "known"
  for m in 1 4 2; do tools/bvh_lab $sc whatif=$m | grep "what-if"; done
done
echo '```'
echo
echo "## One, two or three rays per lane?  The wave schedule simulated on the replay's step sequences (atrium, bounce-1 closest-hit rays, 2 048 rays per wave)"
echo
echo '```'
tools/bvh_lab sponza wavesim=1 | grep "wave sim"
echo '```'
echo
echo "(what-if rows: the hit differs from the device order's on 4 / 9 of 648 000 rays — float-sliver hits that only the device order's extra visits find, DESIGN.md section 2.)"
echo
echo "Reading: 27 % of the node steps hit none of the node's eight children (the ray crossed the node's box but no child's); with the closest distance known in advance the same tree needs"
echo "~10.3-11.3 steps (the bound for ANY visiting order), i.e. the octant order leaves 14 % on the table, of which 8.5 % could be had by skipping popped children that lie beyond the"
echo "best hit (per-child distances on the stack: not affordable, docs/REJECTED.md) and ~5 % by visiting in true distance order.  The builder's splits are not the lever on these"
echo "uniformly tessellated scenes: sweep SAH changes nothing, spatial splits find nothing to separate; on the reference's real model (garage + monke) re-insertion and splits each take 11 %."
