set -x
R=$GRAFT_REPO_ROOT
export PDMK_PLAN_CACHE=$R/gpurun_out/plan_r03g.txt
cd /tmp && export TMPDIR=/tmp
(while sleep 45; do date >> $R/gpurun_out/hb.log; done) &
HB=$!
BARGS="--steps 1 --warmup 1 --no_graph --no_cpu_baseline --no_b16 --no_roofline --no_vae"
rm -rf /tmp/p_f
SECONDS=0
timeout -k 10 800 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -- python3 $R/bench.py $BARGS > $R/gpurun_out/r03g_pmc_f.log 2>&1
echo "rc=$? elapsed=$SECONDS"
kill $HB
ls -la /tmp/p_f/* | head
