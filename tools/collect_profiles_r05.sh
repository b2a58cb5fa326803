# Round-5 evidence in one gpurun call: bench lines for every single-GPU BASELINE config, rocprofv3 kernel stats of the judged
# command (2 utterances in flight) and of the sequential schedule, one utterance's kernel list, the vocoders', the shape tables of
# the kernels built this round (K3f fused pairs, K12 projections), K3f's per-phase stamps, HBM traffic (PMC, separate passes) of
# the roofline kernel, of K3f and of the kNN search.  The PMC summaries are written INTO profiles/ (bench.py reads the JSONs).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r05p; P=$R/gpurun_out/r05p/profiles; mkdir -p $O $P
python3 $R/bench.py --config 2 --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for c in 1 4 5; do python3 $R/bench.py --config $c --steps 20 --warmup 5 --cpu-seconds 3 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; done
python3 $R/bench.py --config 2 --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > $O/bench_cfg2_inflight1.json 2>/dev/null
(cd $R && timeout 300 python3 tools/ab_branch.py 2>&1 | grep -v amdgpu.ids) > $O/ab_branch.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
cp /tmp/pb/b_kernel_stats.csv $O/bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb1 -o b -- python3 $R/bench.py --steps 10 --warmup 3 --inflight 1 --no-cpu-baseline > $O/bench_under_rocprof_inflight1.log 2>&1
cp /tmp/pb1/b_kernel_stats.csv $O/bench_kernel_stats_inflight1.csv
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o t -- python3 $R/tools/profile_pipeline.py > $O/profile_pipeline.log 2>&1
python3 $R/tools/summarize_trace.py /tmp/pp/t_kernel_trace.csv 0 all lastgap > $O/pipeline_kernels.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/d1 -o t -- python3 $R/tools/profile_decoder.py >/dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/d1/t_kernel_trace.csv > $O/decoder_kernels.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/d2 -o t -- python3 $R/tools/profile_decoder.py RefineGAN >/dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/d2/t_kernel_trace.csv > $O/decoder_kernels_refinegan.txt
python3 $R/tools/bench_resblock_bf.py 2>&1 | grep -v amdgpu.ids > $O/rbf_shapes.txt
python3 $R/tools/bench_gemmbf.py 2>&1 | grep -v amdgpu.ids > $O/gemm_shapes.txt
for cfg in "32 3 1" "32 7 1" "32 11 5" "64 3 1" "64 7 1"; do set -- $cfg; C=$1 K=$2 D=$3 python3 $R/tools/stamp_resblock_bf.py 2>&1 | grep -v "amdgpu.ids\|RVC_AMD_LIB"; done > $O/rbf_stamps.txt
# HBM traffic of the roofline kernel: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o f -- python3 $R/tools/pmc_conv.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o w -- python3 $R/tools/pmc_conv.py > /dev/null 2>&1
cp /tmp/pf/f_counter_collection.csv $O/pmc_winobf2_FETCH_SIZE.csv; cp /tmp/pw/w_counter_collection.csv $O/pmc_winobf2_WRITE_SIZE.csv
python3 $R/tools/summarize_pmc.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv $P/pmc_winobf2_k11.json > $O/pmc_winobf2.txt 2>&1
# ... of the fused ResBlock pairs
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/rf -o f -- python3 $R/tools/pmc_rbf.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/rw -o w -- python3 $R/tools/pmc_rbf.py > /dev/null 2>&1
python3 $R/tools/summarize_pmc_rbf.py /tmp/rf/f_counter_collection.csv /tmp/rw/w_counter_collection.csv $P/pmc_resblock_bf.json > $O/pmc_resblock_bf.txt 2>&1
# ... of HuBERT's feature extractor on time-major frames (K13, K12 with a row stride)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/hf -o f -- python3 $R/tools/pmc_hubert_front.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/hw -o w -- python3 $R/tools/pmc_hubert_front.py > /dev/null 2>&1
python3 $R/tools/summarize_pmc_front.py /tmp/hf/f_counter_collection.csv /tmp/hw/w_counter_collection.csv > $O/pmc_hubert_front.txt 2>&1
# SQ counters of the roofline kernel (derived pipe-busy figure next to the raw ones)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pc -o c -- python3 $R/tools/pmc_conv.py > /dev/null 2>&1
python3 - > $O/sq_winobf2.txt <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/pc/c_counter_collection.csv")):
    if "winobf2_conv_kernel" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-40:] + " grid " + r["Grid_Size"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    print(f"{k}  avg {sum(dur[k])/len(dur[k])/1e3:8.1f} us")
    m = {n: sum(v) / len(v) for n, v in d.items()}
    for n, v in sorted(m.items()): print(f"    {n:28s} {v:.4e}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        clk = m['GRBM_GUI_ACTIVE'] / 8 / (sum(dur[k]) / len(dur[k]))          # the counter is summed over the 8 XCDs
        busy = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024)
        print(f"    -> effective clock {clk:.2f} GHz; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = {busy:.3f} of the cycles "
              f"at that clock = {busy * clk / 2.4:.3f} of the 2.4 GHz peak the roofline is quoted against")
PY
# kNN: kernel stats + HBM traffic of one search at 100 k and 2 M rows
(cd $R && bash tools/profile_knn.sh) > $O/knn_profile.log 2>&1
K=$R/gpurun_out/prof_knn
for n in 100000 2000000; do
  python3 $R/tools/summarize_knn_pmc.py $K/knn_${n}_pmc_FETCH_SIZE.csv $K/knn_${n}_pmc_WRITE_SIZE.csv $n $P/pmc_knn_$n.json > $O/knn_pmc_$n.txt 2>&1
  cp $K/knn_${n}_kernels.txt $O/knn_${n}_kernels.txt
done
for c in 2 1 4 5; do python3 - <<PY
import json
d = json.loads([l for l in open("$O/bench_cfg$c.json") if l.startswith("{")][-1])
print("cfg $c:", d["ms_per_step"], "ms/step", d["rtf"], "x RT; host_io", d["host_io"]["ms_per_step"], "; knn", (d.get("roofline_knn") or {}).get("avg_search_ms"), "; dec", (d.get("decoder") or {}).get("ms"), "; cpu", (d.get("cpu_baseline") or {}).get("value"), "; roofline", (d.get("roofline") or {}).get("frac"))
PY
done
python3 -c "
import json; d=json.loads([l for l in open('$O/bench_cfg2_inflight1.json') if l.startswith('{')][-1]); print('cfg 2 inflight 1:', d['ms_per_step'])"
tail -3 $O/pipeline_kernels.txt
