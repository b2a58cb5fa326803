# kernel trace of the judged schedule (2 utterances in flight) for tools/analyze_overlap.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05t
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05t -o t -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rooflines > $R/gpurun_out/r05t/bench.log 2>&1
ls -la $R/gpurun_out/r05t | tail -3
