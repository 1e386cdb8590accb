#!/bin/bash
# The builder / visiting-order experiments as one table (host replay of the device traversal, no GPU): bash tools/bvh_lab_table.sh > profiles/r05_bvh_lab.md
# (round 5: + the HARD stand-ins — the real assets' triangle-size distribution, host/Scenes.h —, + the sizing of a flat test of the level-2 boxes, what-if rows in the product's arithmetic)
cd "$(dirname "$0")/.."
make -s -C tools bvh_lab || exit 1
echo "# Work per ray on the host replay of the device traversal (tools/bvh_lab: the product's builder + csrc/rtx_scene_host.cpp replay_trace), round 5"
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
for sc in sponza sponza_hard bistro bistro_hard garage; do
  row $sc "round-3 builder (16 bins, leaf_stop 2, no re-insertion)" reinsert=0 leaf_stop=2
  row $sc "**default**: 16 bins, split down to single references, 2 re-insertion passes over the <= 200 000 largest nodes" 
  row $sc "full-sweep SAH everywhere, no re-insertion" reinsert=0 sweep=100000000
  row $sc "spatial splits alpha 1e-5, no re-insertion" reinsert=0 split=1e-5
  row $sc "spatial splits + re-insertion" split=1e-5
  row $sc "exact slot assignment, no re-insertion" reinsert=0 slot_assign=1
  row $sc "default with leaf_stop 2 (the default until round 5)" leaf_stop=2
  row $sc "the GPU build's host twin: PLOC radius 16 down to 8 192 clusters (the default was 8 192 when this table was made; 16 384 since), SAH + re-insertion on top" ploc=16
  row $sc "PLOC radius 16 to the root, no re-insertion" ploc=16 ploc_top=1 reinsert=0
  row $sc "default, any-hit NEAREST octant first" any_order=1
  row $sc "default, any-hit FARTHEST octant first" any_order=2
done
echo
echo "## Where the node steps go, and what a better visiting order could save (default builder)"
echo
echo '```'
for sc in sponza sponza_hard bistro bistro_hard; do
  echo "--- $sc"
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
echo "## A flat, wave-uniform test of the <= 64 level-2 boxes instead of the node steps at depths 0 and 1 (VERDICT r04 item 2: sized before building)"
echo
echo '```'
for sc in sponza sponza_hard bistro bistro_hard; do echo "--- $sc"; tools/bvh_lab $sc flat=1 | grep -A3 "flat test"; done
echo '```'
echo
echo "(what-if rows: 0 mismatches.  Round 4's table showed 4 / 9 rays per 648 000 whose hit differed between visiting orders and called them float-sliver hits; they were hits ON AN EDGE —"
echo "u or v exactly 0, u + v one ulp past 1 — decided differently by the tool's own UNFUSED restatement of the triangle test and the replay's fused one.  The tool now uses the"
echo "product's arithmetic (rtx_math.hpp) and the determinant floor of the hit definition: one definition, no order dependence.)"
echo
echo "Reading: 27 % of the node steps hit none of the node's eight children (the ray crossed the node's box but no child's); with the closest distance known in advance the same tree needs"
echo "~10.3-11.3 steps (the bound for ANY visiting order), i.e. the octant order leaves 14 % on the table, of which 8.5 % could be had by skipping popped children that lie beyond the"
echo "best hit (per-child distances on the stack: not affordable, docs/REJECTED.md) and ~5 % by visiting in true distance order.  On the uniformly tessellated stand-ins the builder's"
echo "knobs change nothing (sweep SAH: 0, spatial splits: nothing to separate); on the HARD variants and on the reference's real model (garage + monke) they do: re-insertion"
echo "-13 % / -7 % node steps (atrium / street), spatial splits + re-insertion a further -4 % closest-hit cost (-8 % any-hit on the street, at +30 % references and 7 x the build time)."
