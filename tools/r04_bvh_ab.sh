#!/bin/bash
# Round-4 same-box A/B of BVH builder knobs and the any-hit order on the BVH workloads (kernels timed one after the other).
#   bash tools/r04_bvh_ab.sh > gpurun_out/r04_bvh_ab.log
cd "$(dirname "$0")/.."
run() { echo "## RTX_BVH='$1' $2 $3"; RTX_BVH="$1" timeout -k 10 300 python tools/kernel_ms.py $2 $3 2>&1 | tail -2; }
for rep in 1 2; do
  for S in sponza bistro; do
    run "" $S
    run "reinsert=2,reinsert_frac=0.5" $S
    run "split=1e-5" $S
    run "reinsert=2,reinsert_frac=0.5,split=1e-5" $S
  done
  run "" bistro 28=0
  run "" bistro 28=1
  run "" bistro 28=2
  run "" sponza 28=0
  run "" sponza 28=2
done
