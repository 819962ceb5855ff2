#!/bin/bash
# same-box A/B of bench.py under two environments: tools/ab.sh "<env A>" "<env B>" [rounds] -> img/s and main-step ms per run.
# Runs alternate A B / B A per round: the arm that runs first in a pair measures ~0.4 % faster (two identical binaries, round 3),
# so a fixed order biases sub-percent comparisons.
A="$1"; B="$2"; R="${3:-2}"
for r in $(seq 1 $R); do
  if [ $((r % 2)) = 1 ]; then ORDER="A B"; else ORDER="B A"; fi
  for v in $ORDER; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    out=$(env $E timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_b16 --no_vae --no_roofline 2>/dev/null | tail -1)
    python - "$v" "$E" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print(sys.argv[1], sys.argv[2] or "(default)", "img/s", d["value"], "ms/iter", d["ms_per_step"], "main", d["extras"].get("ms_main_step"), "upper", d["extras"].get("ms_upper_step"), flush=True)
PY
  done
done
