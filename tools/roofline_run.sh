#!/bin/bash
# The counter side of bench.py's roofline, for ONE workload, on the GPU box:
#   1. tools/pmc_run.sh: one rocprofv3 --pmc pass per counter group of `bench.py --workload W` (HBM bytes, VALU instruction classes, lanes, cycles) -> pmc_summary.json
#   2. tools/gen_mix.py: a saturated replay loop of every hot kernel's OWN instruction mix, compiled into tools/valu_peak.hip and run plainly and under
#      --pmc GRBM_GUI_ACTIVE: the instructions per SIMD-cycle that mix reaches on this chip = the kernel's compute peak
#   3. tools/valu_calib.py merges both into <out>/roofline.json (+ .md): per kernel valu instructions per SIMD-cycle achieved / peak = compute_frac (<= 1 by
#      construction), lanes per instruction, HBM bytes per launch; stamped with the hash of the kernel sources so that bench.py can tell a stale profile.
# usage: tools/roofline_run.sh <outdir under gpurun_out/> <workload> [kernel substrings for the replay ...]
out=$1; wl=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p "$out"
bash tools/pmc_run.sh "$out" --workload "$wl" > "$out/pmc_run.log" 2>&1
python3 tools/pmc_summary.py "$out" > "$out/pmc_summary.log" 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize --cuda-device-only -S royaltracer-dx_amd/csrc/rtx_kernels.hip -o "$out/kernels.s" || exit 1
python3 tools/gen_mix.py "$out/pmc_summary.json" "$out/kernels.s" "$@" > "$out/mix_kernels.inc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DRTX_HAVE_MIX -I"$out" tools/valu_peak.hip -o "$out/valu_peak" || exit 1
"$out/valu_peak" calib > "$out/calib_plain.log" 2>&1; echo "calib plain rc=$?"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU -d "$out/calib_g1" -o p --output-format csv -- "$out/valu_peak" calib > "$out/calib_g1.log" 2>&1; echo "calib pmc rc=$?"
rm -f "$out/kernels.s" "$out/valu_peak"
python3 tools/valu_calib.py "$out" "$wl"
