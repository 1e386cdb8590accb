#!/bin/bash
# Same-box A/B of library builds on the headline workload: bash tools/lib_ab_c2.sh <suffix> [<suffix> ...]   ("-" = librtx_hip.so)
D=$PWD/royaltracer-dx_amd
for r in 1 2 3; do for V in "$@"; do [ "$V" = "-" ] && V=""
  RTX_LIB_PATH=$D/librtx_hip$V.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --checksum 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('librtx_hip$V.so', d['config']['workload'], d['ms_per_step'], d.get('accum_sha1'))"
done; done
