# (round 6, historical) A/B of a memory pool reserved up front in bench.py (removed again: no effect) -- kept for the record of the spread across fresh processes
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06s; mkdir -p $O
for i in 1 2 3 4 5; do
  python3 bench.py --no-cpu-baseline --no-rooflines > $O/pool_$i.json 2>/dev/null
  BENCH_NO_POOL=1 python3 bench.py --no-cpu-baseline --no-rooflines > $O/nopool_$i.json 2>/dev/null
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/*pool_*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f.split("/")[-1], d["ms_per_step"], "host_io", d["host_io"]["ms_per_step"], d["timed_region"])
PY
