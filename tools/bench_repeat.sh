# N fresh bench.py processes on one box: the spread of the headline across cold starts, and the utterances-in-flight sweep
# (bash tools/bench_repeat.sh on the GPU box; output gpurun_out/r06r)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06r; mkdir -p $O; cd $R
for i in 1 2 3 4 5 6; do python3 bench.py --no-cpu-baseline --no-rooflines > $O/cfg2_run$i.json 2>/dev/null; done
for m in 2 3 4 3 2 4; do python3 bench.py --inflight $m --steps 24 --warmup 6 --no-cpu-baseline --no-rooflines >> $O/inflight_$m.jsonl 2>/dev/null; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/cfg2_run*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f.split("/")[-1], d["ms_per_step"], "host_io", d["host_io"]["ms_per_step"], d["timed_region"])
for m in (2, 3, 4):
    for l in open("$O/inflight_%d.jsonl" % m):
        if l.startswith("{"):
            d = json.loads(l); print("inflight", m, d["ms_per_step"], "host_io", d["host_io"]["ms_per_step"], d["timed_region"])
PY
