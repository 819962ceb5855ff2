#!/bin/bash
# kernel trace of the timed region of bench.py -> per-kernel time inside one steady-state main step (tools/rocpd_step.py)
# usage (on the GPU box): bash tools/step_trace.sh <tag> [env assignments...]
R=$GRAFT_REPO_ROOT; TAG=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kt_$TAG
env "$@" true
for kv in "$@"; do export "$kv"; done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/p_kt_$TAG -o $TAG -- python3 $R/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_b16 --no_roofline --no_vae > $R/gpurun_out/${TAG}_run.log 2>&1 || exit 8
cd $R
DB=$(find /tmp/p_kt_$TAG -name "*results.db" | head -1)
python tools/rocpd_step.py $DB 5 90 > gpurun_out/${TAG}_step_breakdown.txt 2>&1
head -5 gpurun_out/${TAG}_step_breakdown.txt
python tools/rocpd_gaps.py $DB 5 > gpurun_out/${TAG}_gaps.txt 2>&1
cat gpurun_out/${TAG}_gaps.txt
