#!/bin/bash
# Same-box sweep of one rtx option on the BVH workloads: bash tools/knob_sweep.sh <opt id> <value> [<value> ...]
set -e
O=$1; shift
for V in "$@"; do for WL in sponza_1080p_16spp_8b bistro_1080p_16spp_8b; do
  timeout -k 10 200 python bench.py --workload $WL --steps 5 --warmup 2 --no-extra --no-cpu-baseline --checksum --opt $O=$V 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('opt $O=$V', d['config']['workload'], d['ms_per_step'], d.get('accum_sha1'))"
done; done
