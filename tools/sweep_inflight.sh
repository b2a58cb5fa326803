#!/bin/bash
# throughput of the judged workload against the number of utterances in flight
mkdir -p gpurun_out/inflight
for n in ${NS:-1 2 3 4}; do
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-rooflines --inflight $n > gpurun_out/inflight/n$n.json 2> gpurun_out/inflight/n$n.err
  python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/inflight/n$n.json") if x.startswith("{")][-1])
print("inflight $n:", l["ms_per_step"], "ms/utt;", "host_io", l["host_io"]["ms_per_step"])
PY
done
