#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and wave-cycle breakdown from one rocprofv3 PMC pass
(`--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY --output-format csv` of `python3 bench.py --steps 1 --warmup 1 --no_graph ...`).
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)   (GUI_ACTIVE is summed over the 8 XCDs,
                MI355X_MICROARCH.md "DVFS give-back"; MFMA_BUSY counts cycles per SIMD, summed over the chip)
  parked / issue_stall / issuing = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint buckets)
usage: tools/summarize_sq.py <pmc_dir> <out.json>"""
import collections, csv, glob, json, re, sys


def symbol(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n).strip()


agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        s = symbol(r["Kernel_Name"])
        agg[s][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (s, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            cnt[s] += 1
out = {}
for s, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if gui <= 0 or wc <= 0:
        continue
    out[s] = {"launches_profiled": cnt[s], "gui_active_cycles_per_launch": gui / 8 / cnt[s],
              "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * gui / 8),
              "waves_parked": c.get("SQ_WAIT_ANY", 0.0) / wc, "waves_issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
              "waves_issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for s, v in sorted(out.items(), key=lambda kv: -kv[1]["gui_active_cycles_per_launch"] * kv[1]["launches_profiled"])[:24]:
    print(f"{s[:66]:66s} n={v['launches_profiled']:5d} mfma {100 * v['mfma_util']:5.1f}%  parked {100 * v['waves_parked']:5.1f}%  "
          f"stall {100 * v['waves_issue_stalled']:5.1f}%  issue {100 * v['waves_issuing']:5.1f}%")
