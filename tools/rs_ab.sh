#!/bin/bash
# Same-box A/B of library builds on the ReSTIR frame: bash tools/rs_ab.sh <suffix> [<suffix> ...]   ("-" = librtx_hip.so)
D=$PWD/royaltracer-dx_amd
for r in 1 2 3; do for V in "$@"; do [ "$V" = "-" ] && V=""
  echo -n "librtx_hip$V.so: "; RTX_LIB_PATH=$D/librtx_hip$V.so python3 tools/restir_time.py sponza garage frames=6 2>/dev/null | grep "frame 5" | awk '{printf "%s %s ms  ", $1, $5} END {print ""}'
done; done
