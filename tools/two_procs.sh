#!/bin/bash
# two independent bench processes on the one GPU: is a single process bound by its own host side (GIL) or by the GPU?
mkdir -p gpurun_out/tp
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-rooflines --inflight 2 > gpurun_out/tp/a.json 2>/dev/null &
PA=$!
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-rooflines --inflight 2 > gpurun_out/tp/b.json 2>/dev/null &
PB=$!
wait $PA $PB
python - <<'PY'
import json
for n in "ab":
    l = json.loads([x for x in open(f"gpurun_out/tp/{n}.json") if x.startswith("{")][-1])
    print("process", n, l["ms_per_step"], "ms/utt")
PY
