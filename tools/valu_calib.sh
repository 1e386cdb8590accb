#!/bin/bash
# Calibration of the compute roofline on the GPU box: builds tools/valu_peak.hip there, runs its `calib` mode once plainly (the kernels' own cycle measurements)
# and once per counter group under rocprofv3 --pmc.  Summarise with: python3 tools/valu_calib.py <outdir>   (writes profiles/r03_valu_calib.{md,json})
# usage: tools/valu_calib.sh <outdir under gpurun_out/>
out=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/valu_peak.hip -o "$out/valu_peak" || exit 1
"$out/valu_peak" calib > "$out/plain.log" 2>&1; echo "plain rc=$?"
i=0
for grp in "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d "$out/g$i" -o p --output-format csv -- "$out/valu_peak" calib > "$out/g$i.log" 2>&1
  echo "group $i rc=$?"
done
