# Round-3 evidence in one gpurun call: bench lines for every single-GPU BASELINE config, rocprofv3 kernel stats of the
# judged command (2 utterances in flight) and of the sequential schedule, one utterance's kernel list, the vocoder's,
# the conv-shape table of the three ResBlock conv forms, the winobf ablation.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r03; mkdir -p $O
python3 $R/bench.py --config 2 --steps 8 --warmup 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for c in 1 4 5; do python3 $R/bench.py --config $c --steps 8 --warmup 2 --cpu-seconds 3 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; done
python3 $R/bench.py --config 2 --steps 8 --warmup 2 --inflight 1 --no-cpu-baseline > $O/bench_cfg2_inflight1.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
cp /tmp/pb/b_kernel_stats.csv $O/bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb1 -o b -- python3 $R/bench.py --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline > $O/bench_under_rocprof_inflight1.log 2>&1
cp /tmp/pb1/b_kernel_stats.csv $O/bench_kernel_stats_inflight1.csv
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o t -- python3 $R/tools/profile_pipeline.py > $O/profile_pipeline.log 2>&1
python3 $R/tools/summarize_trace.py /tmp/pp/t_kernel_trace.csv 0 all lastgap > $O/pipeline_kernels.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/d1 -o t -- python3 $R/tools/profile_decoder.py >/dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/d1/t_kernel_trace.csv > $O/decoder_kernels_nsf.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/d2 -o t -- python3 $R/tools/profile_decoder.py RefineGAN >/dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/d2/t_kernel_trace.csv > $O/decoder_kernels_refinegan.txt
python3 $R/tools/bench_convbf.py > $O/convbf_shapes.txt 2>&1
(cd $R && bash tools/ablate_winobf.sh) > $O/winobf_ablation.txt 2>&1
for c in 2 1 4 5; do python3 - <<PY
import json
d = json.loads([l for l in open("$O/bench_cfg$c.json") if l.startswith("{")][-1])
print("cfg $c:", d["ms_per_step"], "ms/step", d["rtf"], "x RT; host_io", d["host_io"]["ms_per_step"], "; knn", (d.get("roofline_knn") or {}).get("avg_search_ms"), "; dec", (d.get("decoder") or {}).get("ms"), "; cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
python3 -c "
import json; d=json.loads([l for l in open('$O/bench_cfg2_inflight1.json') if l.startswith('{')][-1]); print('cfg 2 inflight 1:', d['ms_per_step'])"
tail -3 $O/pipeline_kernels.txt
