#!/bin/bash
# Everything bench.py and DESIGN.md quote from counters / traces, recorded in one go on the GPU box (run through gpurun; ~10 minutes):
#   roofline counters + instruction-mix replay of the three BASELINE workloads, kernel traces of the same, kernel trace + counters of the ReSTIR frame.
# usage: tools/record_round.sh <tag, e.g. r03>      -> gpurun_out/<tag>_*/   (copy the summaries to profiles/ with tools/collect_profiles.py)
tag=$1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
bash tools/roofline_run.sh gpurun_out/${tag}_roof_cornell cornell_1080p_64spp_8b "k_bounce_small<4, false, true, true>" "k_bounce_small<4, true, true, false>" "k_raygen_trace_small" > gpurun_out/${tag}_roof_cornell.log 2>&1; echo "cornell roofline rc=$?"
bash tools/roofline_run.sh gpurun_out/${tag}_roof_sponza sponza_1080p_16spp_8b "k_trace_closest<0, false, 6>" "k_trace_shadow<0, false, 6, 0>" "k_shade<false, true>" > gpurun_out/${tag}_roof_sponza.log 2>&1; echo "sponza roofline rc=$?"
bash tools/roofline_run.sh gpurun_out/${tag}_roof_bistro bistro_1080p_16spp_8b "k_trace_closest<0, false, 6>" "k_trace_shadow<0, false, 6, 0>" "k_shade<false, false>" > gpurun_out/${tag}_roof_bistro.log 2>&1; echo "bistro roofline rc=$?"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in cornell_1080p_64spp_8b sponza_1080p_16spp_8b bistro_1080p_16spp_8b; do
  d=gpurun_out/${tag}_kt_${wl%%_*}; mkdir -p $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d -o p --output-format csv -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $d/bench.log 2>&1; echo "kernel trace $wl rc=$?"
done
for sc in garage sponza bistro; do
  d=gpurun_out/${tag}_kt_restir_$sc; mkdir -p $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $d -o p --output-format csv -- python3 tools/restir_time.py $sc frames=4 > $d/run.log 2>&1; echo "kernel trace restir $sc rc=$?"
done
PMC_CMD="tools/restir_time.py sponza frames=2" bash tools/pmc_run.sh gpurun_out/${tag}_pmc_restir_sponza > /dev/null 2>&1; python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_restir_sponza > /dev/null; echo "restir pmc rc=$?"
