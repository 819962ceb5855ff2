#!/bin/bash
# SQ counters of the attention kernels at N = 4096 (two --pmc passes; run on the GPU box): tools/attn_pmc.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-attn}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa1 /tmp/pa2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pa1 -- python3 $R/tools/attn_pmc.py > $R/gpurun_out/${TAG}_pmc1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVES --output-format csv -d /tmp/pa2 -- python3 $R/tools/attn_pmc.py > $R/gpurun_out/${TAG}_pmc2.log 2>&1 || exit 2
cd $R
python tools/summarize_sq.py /tmp/pa1 gpurun_out/${TAG}_sq.json | grep attn
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("/tmp/pa2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" not in r["Kernel_Name"]: continue
        s = r["Kernel_Name"].split("(")[0][-60:]
        agg[s][r["Counter_Name"]] += float(r["Counter_Value"])
for s, c in agg.items():
    print(s, {k_: f"{v:.3g}" for k_, v in c.items()})
PY
