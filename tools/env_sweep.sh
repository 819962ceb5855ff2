#!/bin/bash
# same-box sweep of bench.py under several environments (tools/ab.sh conventions): tools/env_sweep.sh ROUNDS "<env 1>" "<env 2>" ...
# "-" = the default environment.  Every round runs all of them, the order reversed on even rounds.  SWEEP_ARGS="--no_graph" adds
# bench.py arguments to every run.
R="$1"; shift
N=$#
for r in $(seq 1 $R); do
  idx=$(seq 1 $N); [ $((r % 2)) = 0 ] && idx=$(seq $N -1 1)
  for i in $idx; do
    E="${!i}"; [ "$E" = "-" ] && E=""
    out=$(env $E timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_b16 --no_vae --no_roofline $SWEEP_ARGS 2>/dev/null | tail -1)
    python - "${E:-(default)}" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(sys.argv[1], "img/s", d["value"], "ms/iter", d["ms_per_step"], "main", d["extras"].get("ms_main_step"), "upper", d["extras"].get("ms_upper_step"), flush=True)
PY
  done
done
