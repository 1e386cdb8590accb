#!/bin/bash
# Sweep of the pending-triangle depth (make PEND=3/4 builds) x vote weights (trace_sched 5-7): frame time, checksum and lane utilisation.
# usage (GPU box): bash tools/pend_sweep.sh > gpurun_out/pend_sweep.log
set -e
D=royaltracer-dx_amd
for S in 6 5 7; do for V in "" _p3 _p4; do
  echo "=== lib librtx_hip$V.so sched $S"
  for WL in sponza_1080p_16spp_8b bistro_1080p_16spp_8b; do
    RTX_LIB_PATH=$PWD/$D/librtx_hip$V.so timeout -k 10 200 python bench.py --workload $WL --steps 5 --warmup 2 --no-extra --no-cpu-baseline --checksum --opt 10=$S 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['workload'], d['ms_per_step'], d.get('accum_sha1'))"
  done
  RTX_LIB_PATH=$PWD/$D/librtx_hip_prof$V.so timeout -k 10 200 python tools/traversal_profile.py sponza $S 2>/dev/null | grep -v "^Exception\|AttributeError\|Traceback\|File " || true
done; done
