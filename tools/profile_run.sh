#!/bin/bash
# rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE passes for one bench.py workload, summarised into profiles/.
# Usage: tools/profile_run.sh <out_prefix under gpurun_out/> <summary note> <bench.py args...>
pre=$1; note=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
d=gpurun_out/$pre; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --stats -d $d/kt -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra "$@" > $d/kt.log 2>&1; echo "kt rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $d/fetch -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > $d/fetch.log 2>&1; echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $d/write -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > $d/write.log 2>&1; echo "write rc=$?"
python3 tools/rocprof_summary.py $d/kt $d/fetch $d/write $d/summary "$note"
cp $d/kt/*kernel_stats.csv $d/summary_kernel_stats.csv 2>/dev/null
tail -1 $d/kt.log
