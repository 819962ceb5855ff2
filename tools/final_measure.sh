set -x
R=$GRAFT_REPO_ROOT
export PDMK_PLAN_CACHE=$R/gpurun_out/plan_r04h.txt
rm -f $PDMK_PLAN_CACHE
cd $R
(while sleep 45; do date >> $R/gpurun_out/hb.log; done) &
# 1. fill the plan cache (no profiler), short
timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_b16 --no_roofline --no_vae > gpurun_out/r04h_fill.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 1 --warmup 1 --no_graph --no_cpu_baseline --no_b16 --no_roofline --no_vae"
rm -rf /tmp/p_f /tmp/p_w /tmp/p_s /tmp/p_kt
timeout -k 10 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -- python3 $R/bench.py $BARGS > $R/gpurun_out/r04h_pmc_f.log 2>&1 || exit 2
timeout -k 10 700 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -- python3 $R/bench.py $BARGS > $R/gpurun_out/r04h_pmc_w.log 2>&1 || exit 3
timeout -k 10 700 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/p_s -- python3 $R/bench.py $BARGS > $R/gpurun_out/r04h_pmc_s.log 2>&1 || exit 4
cd $R
(while sleep 45; do date >> $R/gpurun_out/hb.log; done) &
python tools/summarize_pmc.py /tmp/p_f /tmp/p_w gpurun_out/r04h_pmc_hbm_traffic.json > gpurun_out/r04h_pmc_hbm_traffic.txt 2>&1 || exit 5
python tools/summarize_sq.py /tmp/p_s gpurun_out/r04h_pmc_sq.json > gpurun_out/r04h_pmc_sq.txt 2>&1 || exit 6
cp gpurun_out/r04h_pmc_hbm_traffic.json profiles/r04_pmc_hbm_traffic.json
# 2. the bench line
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04h_bench.json 2> gpurun_out/r04h_bench.err || exit 7
# 3. kernel trace of the timed region
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o r04h -- python3 $R/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_b16 --no_roofline --no_vae > $R/gpurun_out/r04h_prof_run.log 2>&1 || exit 8
cd $R
(while sleep 45; do date >> $R/gpurun_out/hb.log; done) &
DB=$(find /tmp/p_kt -name "*results.db" | head -1)
python tools/rocpd_step.py $DB 5 80 > gpurun_out/r04h_step_breakdown.txt 2>&1
python tools/rocpd_stats.py $DB > gpurun_out/r04h_bench_kernel_stats.csv 2>&1
head -12 gpurun_out/r04h_step_breakdown.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
PY
# 4. everything the round's DESIGN.md quotes lands under profiles/ with the round's prefix
cp gpurun_out/r04h_bench.json profiles/r04_bench.json
cp gpurun_out/r04h_step_breakdown.txt profiles/r04_step_breakdown.txt
cp gpurun_out/r04h_bench_kernel_stats.csv profiles/r04_bench_kernel_stats.csv
cp gpurun_out/r04h_pmc_sq.json profiles/r04_pmc_sq.json
cp profiles/r04_pmc_hbm_traffic.json gpurun_out/r04h_pmc_hbm_traffic_copy.json
