#!/bin/bash
# A/B of the vocoder conv selection: RVC_WINO=0 (direct only), 1 (policy), 2 (Winograd wherever supported)
mkdir -p gpurun_out/ab
for w in 0 1 2; do
  RVC_WINO=$w python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ab/wino$w.json 2> gpurun_out/ab/wino$w.err
  echo "WINO=$w rc=$?"; tail -3 gpurun_out/ab/wino$w.err
  python - <<PY
import json
try:
    l = json.loads(open("gpurun_out/ab/wino$w.json").read().strip().splitlines()[-1])
    print(l["ms_per_step"], l["value"], {k: l[k] for k in l if k.startswith("roofline")}, l.get("sequential"))
except Exception as e:
    print("no line:", e)
PY
done
