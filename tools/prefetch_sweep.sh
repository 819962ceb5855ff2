#!/bin/bash
# same-box sweep of the teacher prefetch: off, and queued behind backward graph 0 / 2 / 4 / 5 (tools/ab.sh conventions)
for v in "PDMK_TEACHER_PREFETCH=0" "PDMK_PREFETCH_AT=0" "PDMK_PREFETCH_AT=2" "PDMK_PREFETCH_AT=4" "PDMK_PREFETCH_AT=5" "PDMK_TEACHER_PREFETCH=0" "PDMK_PREFETCH_AT=0"; do
  out=$(env $v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_b16 --no_vae --no_roofline 2>/dev/null | tail -1)
  python - "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(sys.argv[1], "img/s", d["value"], "ms/iter", d["ms_per_step"], "main", d["extras"].get("ms_main_step"), "upper", d["extras"].get("ms_upper_step"), flush=True)
PY
done
