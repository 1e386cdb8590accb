#!/bin/bash
# Collect SQ / TCC counters for one bench.py invocation: one rocprofv3 --pmc pass per counter group (no trace domains
# combined with --pmc), 1 frame each.  Usage: tools/pmc_run.sh <outdir> <bench.py args...>
# (FETCH_SIZE and WRITE_SIZE in passes of their own: together they abort rocprofv3 on this image)
# PMC_GROUPS="1 2 3" restricts the passes.  PMC_CMD="tools/restir_time.py sponza frames=2" profiles another python script than bench.py (its arguments instead of
# the bench arguments).  Summarise with: python3 tools/pmc_summary.py <outdir>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p "$out"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  if [ -n "$PMC_GROUPS" ] && [[ " $PMC_GROUPS " != *" $i "* ]]; then continue; fi
  if [ -n "$PMC_CMD" ]; then
    timeout -k 10 300 rocprofv3 --pmc $grp -d "$out/g$i" -o p --output-format csv -- python3 $PMC_CMD > "$out/g$i.log" 2>&1
  else
    timeout -k 10 300 rocprofv3 --pmc $grp -d "$out/g$i" -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > "$out/g$i.log" 2>&1
  fi
  echo "group $i rc=$?"
done
