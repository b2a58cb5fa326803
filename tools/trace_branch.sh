cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05b
for n in 0 2; do
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05b/trace_s$n -o t -- python3 $R/bench.py --inflight 1 --steps 3 --warmup 2 --no-cpu-baseline --no-rooflines --vocoder-side-streams $n > $R/gpurun_out/r05b/trace_s$n.json 2> $R/gpurun_out/r05b/trace_s$n.err
done
ls -la $R/gpurun_out/r05b/trace_s0/* | head
