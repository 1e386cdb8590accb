#!/bin/bash
# Same-box A/B of library builds on the BVH workloads: bash tools/lib_ab.sh <suffix> [<suffix> ...]   ("-" = librtx_hip.so, "_head" = librtx_hip_head.so, ...)
D=$PWD/royaltracer-dx_amd
for r in 1 2; do for V in "$@"; do [ "$V" = "-" ] && V=""; for S in sponza bistro; do
  RTX_LIB_PATH=$D/librtx_hip$V.so timeout -k 10 200 python tools/kernel_ms.py $S 2>/dev/null | tail -1
done; done; done
