#!/bin/bash
# sweep of the grouped weight-gradient launch geometry (bench.py main-step ms per setting, two passes in alternating order)
for pass in 1 2; do
  if [ $pass = 1 ]; then L="512:32 768:32 1024:32 512:16 512:64 384:32"; else L="384:32 512:64 512:16 1024:32 768:32 512:32"; fi
  for cfg in $L; do
    T=${cfg%%:*}; K=${cfg##*:}
    out=$(PDMK_WG_TARGET=$T PDMK_WG_MINK=$K timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_b16 --no_vae --no_roofline 2>/dev/null | tail -1)
    python - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("target:mink", sys.argv[1], "img/s", d["value"], "main", d["extras"].get("ms_main_step"), "upper", d["extras"].get("ms_upper_step"), flush=True)
PY
  done
done
