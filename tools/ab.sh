#!/bin/bash
# same-box A/B of bench.py under two environments: tools/ab.sh "<env A>" "<env B>" [rounds] -> img/s and main-step ms per run
A="$1"; B="$2"; R="${3:-2}"
for r in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    out=$(env $E timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_b16 --no_vae --no_roofline 2>/dev/null | tail -1)
    python - "$v" "$E" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print(sys.argv[1], sys.argv[2] or "(default)", "img/s", d["value"], "ms/iter", d["ms_per_step"], "main", d["extras"].get("ms_main_step"), "upper", d["extras"].get("ms_upper_step"), flush=True)
PY
  done
done
