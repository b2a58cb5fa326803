cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o t -- python3 $R/tools/profile_pipeline.py > $O/profile_pipeline.log 2>&1
python3 $R/tools/summarize_trace.py /tmp/pp/t_kernel_trace.csv 0 all lastgap > $O/pipeline_kernels.txt
cd $R
for i in 1 2; do
for c in 2 1; do
timeout 300 python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-rooflines > $O/b_c${c}_$i.json 2> $O/b_c${c}_$i.err; echo "cfg $c $(grep -o '"ms_per_step": [0-9.]*' $O/b_c${c}_$i.json | head -1)"
done; done
timeout 300 python3 bench.py --inflight 1 --steps 8 --warmup 2 --no-cpu-baseline --no-rooflines > $O/b_i1.json 2> $O/b_i1.err; echo "inflight 1 $(grep -o '"ms_per_step": [0-9.]*' $O/b_i1.json | head -1)"
