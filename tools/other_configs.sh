#!/bin/bash
# one-GPU numbers for the other BASELINE.json configurations (parity-test cases, not the bench line): same bilevel cadence
F="--steps 10 --warmup 3 --no_cpu_baseline --no_b16 --no_roofline --no_vae"
for cfg in "--batch 8 --budget 0.55" "--batch 16 --budget 0.55" "--batch 8 --budget 0.82" "--batch 8 --budget 0.18" "--batch 4 --budget 1.0 --latent 96"; do
  out=$(timeout -k 10 280 python bench.py $F $cfg 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
e = d["extras"]
print(f"{sys.argv[1]:40s} img/s {d['value']:8.2f}  ms/iter {d['ms_per_step']:7.2f}  main {e.get('ms_main_step')}  upper {e.get('ms_upper_step')}  model TFLOP/s {e.get('model_tflops_per_gpu')}", flush=True)
PY
done
