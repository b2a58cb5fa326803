#!/bin/bash
# round 4: full GPU suite on the winobf2 tree + bench lines of cfg 2 / 4 / 5
mkdir -p gpurun_out/r04
timeout 3000 python -m pytest tests/ -m gpu -q -x > gpurun_out/r04/gpu_suite_a.txt 2>&1
tail -5 gpurun_out/r04/gpu_suite_a.txt
timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_cfg2_a.json 2> gpurun_out/r04/bench_cfg2_a.err; tail -c 600 gpurun_out/r04/bench_cfg2_a.err
timeout 400 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_cfg4_a.json 2> gpurun_out/r04/bench_cfg4_a.err
timeout 400 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_cfg5_a.json 2> gpurun_out/r04/bench_cfg5_a.err
timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --inflight 1 --no-rooflines > gpurun_out/r04/bench_cfg2_inflight1_a.json 2> /dev/null
python - <<'PY'
import json
for n in ("cfg2_start", "cfg4_start", "cfg2_a", "cfg4_a", "cfg5_a", "cfg2_inflight1_a"):
    try:
        d = json.loads(open(f"gpurun_out/r04/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], "ms/step", "decoder", (d.get("decoder") or {}).get("ms"), "roofline", (d.get("roofline") or {}).get("avg_launch_ms"), (d.get("roofline") or {}).get("frac"), "knn", (d.get("roofline_knn") or {}).get("ms"))
    except Exception as e:
        print(n, "failed", e)
PY
