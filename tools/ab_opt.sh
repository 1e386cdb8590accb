#!/bin/bash
# Same-box alternating A/B of one rtx option on the BVH workloads: bash tools/ab_opt.sh <opt id> <value A> <value B> [rounds] [extra bench args]
set -e
O=$1; A=$2; B=$3; R=${4:-3}; shift 4 || true
for r in $(seq $R); do for V in $A $B; do for WL in sponza_1080p_16spp_8b bistro_1080p_16spp_8b; do
  timeout -k 10 200 python bench.py --workload $WL --steps 5 --warmup 2 --no-extra --no-cpu-baseline --checksum --opt $O=$V "$@" 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('opt $O=$V', d['config']['workload'], d['ms_per_step'], d.get('accum_sha1'), {k: round(v, 2) for k, v in d.get('kernel_ms_per_step', {}).items()} if isinstance(d.get('kernel_ms_per_step'), dict) else '')"
done; done; done
