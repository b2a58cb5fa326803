R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/b_rl.json 2> $O/b_rl.err
python3 - <<PY
import json
d=json.loads([l for l in open("$O/b_rl.json") if l.startswith("{")][-1]); r=d["roofline"]
print(d["ms_per_step"], r["avg_launch_ms"], r["batch_launch_ms"], r["frac"])
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb1 -o b -- python3 $R/bench.py --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline > $O/rl_under_rocprof.log 2>&1
grep "winobf2_conv_kernel<11" /tmp/pb1/b_kernel_stats.csv | cut -c1-160
python3 - <<PY
import json
d=json.loads([l for l in open("$O/rl_under_rocprof.log") if l.startswith("{")][-1]); r=d["roofline"]
print("under rocprof:", d["ms_per_step"], r["avg_launch_ms"], r["batch_launch_ms"], r["frac"])
PY
