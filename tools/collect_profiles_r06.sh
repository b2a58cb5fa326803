# Round-6 evidence in one gpurun call (bash tools/collect_profiles_r06.sh on the GPU box; results under gpurun_out/r06p, the PMC JSONs
# bench.py reads under gpurun_out/r06p/profiles -> copy both into profiles/):
#   * bench lines of every single-GPU BASELINE config on ONE box (cfg 4 against cfg 2 is an A/B: verdict item 1), the sequential
#     schedule, a > 41 s clip (100 s: the long-input front, verdict item 7);
#   * rocprofv3 kernel stats of the judged command and of the sequential schedule; one utterance's kernel list; the vocoders';
#   * K3f with one-term taps: shape table, SQ counters (matrix pipe busy), HBM traffic; K10b: shape table, kernel durations, stamps;
#   * the roofline kernel (K3y): HBM traffic, SQ counters, the L2 -> CU request counters the verdict named;
#   * kNN: the screened search and the STREAMING regime (32 queries), kernel stats + HBM traffic (verdict item 8);
#   * tools/micro/l2_ingest: what one CU can pull out of L2 next to a matrix stream (the "ingest wall" of DESIGN section 4, measured).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r06p; P=$R/gpurun_out/r06p/profiles; mkdir -p $O $P
cp $R/profiles/pmc_*.json $P/ 2>/dev/null
python3 $R/bench.py --config 2 --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python3 $R/bench.py --config 4 --steps 20 --warmup 5 --cpu-seconds 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python3 $R/bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2_again.json 2>/dev/null
python3 $R/bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_again.json 2>/dev/null
for c in 1 5; do python3 $R/bench.py --config $c --steps 20 --warmup 5 --cpu-seconds 3 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; done
python3 $R/bench.py --config 2 --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > $O/bench_cfg2_inflight1.json 2>/dev/null
python3 $R/bench.py --config 2 --steps 20 --warmup 5 --inflight 2 --no-cpu-baseline --no-rooflines > $O/bench_cfg2_inflight2.json 2>/dev/null
python3 $R/bench.py --config 2 --seconds 100 --steps 8 --warmup 2 --no-cpu-baseline --no-rooflines > $O/bench_cfg2_100s.json 2> $O/bench_cfg2_100s.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
cp /tmp/pb/b_kernel_stats.csv $O/bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb1 -o b -- python3 $R/bench.py --steps 10 --warmup 3 --inflight 1 --no-cpu-baseline > $O/bench_under_rocprof_inflight1.log 2>&1
cp /tmp/pb1/b_kernel_stats.csv $O/bench_kernel_stats_inflight1.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb4 -o b -- python3 $R/bench.py --config 4 --steps 10 --warmup 3 --inflight 1 --no-cpu-baseline > $O/bench_cfg4_under_rocprof_inflight1.log 2>&1
cp /tmp/pb4/b_kernel_stats.csv $O/bench_cfg4_kernel_stats_inflight1.csv
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o t -- python3 $R/tools/profile_pipeline.py > $O/profile_pipeline.log 2>&1
python3 $R/tools/summarize_trace.py /tmp/pp/t_kernel_trace.csv 0 all lastgap > $O/pipeline_kernels.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/d1 -o t -- python3 $R/tools/profile_decoder.py >/dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/d1/t_kernel_trace.csv > $O/decoder_kernels.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/d2 -o t -- python3 $R/tools/profile_decoder.py RefineGAN >/dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/d2/t_kernel_trace.csv > $O/decoder_kernels_refinegan.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/d3 -o t -- python3 $R/tools/profile_decoder.py "MRF HiFi-GAN" bf16 >/dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/d3/t_kernel_trace.csv > $O/decoder_kernels_mrf_bf16.txt
BENCH_ONE=1 python3 $R/tools/bench_resblock_bf.py 2>&1 | grep -v amdgpu.ids > $O/rbf1_shapes.txt
python3 $R/tools/bench_convbf1.py 2>&1 | grep -v amdgpu.ids > $O/convbf1_shapes.txt
# K10b (RMVPE's 3x3 convs as bf16x3 products) against K10: event timings per shape, the kernels' own durations, the per-phase stamps
python3 $R/tools/bench_conv2dbf.py 2>&1 | grep -v amdgpu.ids > $O/conv2dbf_shapes.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/c2 -o t -- python3 $R/tools/bench_conv2dbf.py > /dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/c2/t_kernel_trace.csv > $O/conv2dbf_kernels.txt
(echo "tools/stamp_conv2dbf.py (ablation build, RVC_C2B_DEBUG=64): cycle stamps of compute wave 0 and stager wave 4 of three workgroups per shape; 108 matrix instructions of 32 cycles = 3456 cycles per item"; cd $R && RVC_AMD_LIB=$R/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so RVC_C2B_DEBUG=64 python3 tools/stamp_conv2dbf.py 2>&1 | grep -v "amdgpu.ids\|RVC_AMD_LIB") > $O/conv2dbf_stamps.txt
# HBM traffic of the roofline kernel: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o f -- python3 $R/tools/pmc_conv.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o w -- python3 $R/tools/pmc_conv.py > /dev/null 2>&1
python3 $R/tools/summarize_pmc.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv $P/pmc_winobf2_k11.json > $O/pmc_winobf2.txt 2>&1
# ... of the fused ResBlock pairs with one-term taps (cfg 4)
ONE=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/rf -o f -- python3 $R/tools/pmc_rbf.py > /dev/null 2>&1
ONE=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/rw -o w -- python3 $R/tools/pmc_rbf.py > /dev/null 2>&1
python3 $R/tools/summarize_pmc_rbf.py /tmp/rf/f_counter_collection.csv /tmp/rw/w_counter_collection.csv $P/pmc_resblock_bf1.json > $O/pmc_resblock_bf1.txt 2>&1
# ... of K10b per U-Net level
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/c2f -o f -- python3 $R/tools/pmc_conv2dbf.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/c2w -o w -- python3 $R/tools/pmc_conv2dbf.py > /dev/null 2>&1
python3 $R/tools/summarize_pmc_conv2dbf.py /tmp/c2f/f_counter_collection.csv /tmp/c2w/w_counter_collection.csv > $O/pmc_conv2dbf.txt 2>&1
# SQ counters: the roofline kernel, then the one-term pairs (derived pipe-busy figure next to the raw ones)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pc -o c -- python3 $R/tools/pmc_conv.py > /dev/null 2>&1
ONE=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pc1 -o c -- python3 $R/tools/pmc_rbf.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pc2 -o c -- python3 $R/tools/pmc_convbf1.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cf -o f -- python3 $R/tools/pmc_convbf1.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cw -o w -- python3 $R/tools/pmc_convbf1.py > /dev/null 2>&1
python3 - > $O/pmc_convbf1.txt <<'PY'
import csv, collections, statistics as st
def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d.setdefault((r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"]), []).append(float(r["Counter_Value"]) * 1024.0)
    return d
f, w = load("/tmp/cf/f_counter_collection.csv"), load("/tmp/cw/w_counter_collection.csv")
for (name, grid), v in f.items():
    if "convbf1_kernel" in name or "copyBuffer" in name:
        C = 256 if "256>" in name else 128
        L = 38376 if C == 256 else 383760
        t = C * L * 4 / 1e6
        print(f"{name:40s} launches {len(v):3d}: FETCH_SIZE raw {st.mean(v)/1e6:7.1f} MB (x2 for 16-byte / x1.77 for 4-byte reads), WRITE_SIZE {st.mean(w.get((name, grid), [0]))/1e6:7.1f} MB"
              + (f"; algorithmic: x {t:.1f} + res {t:.1f} read, y {t:.1f} written" if "convbf1" in name else "  (calibration: a tensor copy)"))
PY
# the L2 -> CU request path and the LDS of the roofline kernel (verdict item 3: "write the bound down with counters")
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d /tmp/pt -o c -- python3 $R/tools/pmc_conv.py > $O/pmc_tcp.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pl -o c -- python3 $R/tools/pmc_conv.py > $O/pmc_lds.log 2>&1
python3 - > $O/sq_counters.txt <<'PY'
import csv, collections, os
def table(path, want):
    if not os.path.exists(path): print("missing", path); return
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if want in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][-40:] + " grid " + r["Grid_Size"]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, d in agg.items():
        n = max(len(v) for v in d.values())
        print(f"{k}  avg {sum(dur[k])/len(dur[k])/1e3:8.1f} us")
        m = {nm: sum(v) / len(v) for nm, v in d.items()}
        for nm, v in sorted(m.items()): print(f"    {nm:32s} {v:.4e}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            clk = m['GRBM_GUI_ACTIVE'] / 8 / (sum(dur[k]) / len(dur[k]))          # the counter is summed over the 8 XCDs
            busy = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024)
            print(f"    -> effective clock {clk:.2f} GHz; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = {busy:.3f} of the cycles "
                  f"at that clock = {busy * clk / 2.4:.3f} of the 2.4 GHz peak")
        if "TCP_TCC_READ_REQ_sum" in m:
            cyc = sum(dur[k]) / len(dur[k]) * 1.95          # ns x ~1.95 GHz (the SQ pass's effective clock for this kernel)
            print(f"    -> L1 -> L2 read requests {m['TCP_TCC_READ_REQ_sum']:.3e} x 64 B = {m['TCP_TCC_READ_REQ_sum']*64/1e6:.0f} MB per launch = "
                  f"{m['TCP_TCC_READ_REQ_sum']*64/256/cyc:.1f} B / cycle / CU at 1.95 GHz")
        if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
            print(f"    -> LDS bank-conflict cycles / LDS active cycles = {m['SQ_LDS_BANK_CONFLICT']/m['SQ_LDS_IDX_ACTIVE']:.3f}")
print("==== winobf2_conv_kernel (K3y): SQ pass"); table("/tmp/pc/c_counter_collection.csv", "winobf2_conv_kernel")
print("==== winobf2_conv_kernel (K3y): L1 -> L2 requests"); table("/tmp/pt/c_counter_collection.csv", "winobf2_conv_kernel")
print("==== winobf2_conv_kernel (K3y): LDS / instruction mix"); table("/tmp/pl/c_counter_collection.csv", "winobf2_conv_kernel")
print("==== resblock_bf_kernel with one-term taps (K3f, cfg 4): SQ pass"); table("/tmp/pc1/c_counter_collection.csv", "resblock_bf_kernel")
print("==== convbf1_kernel (K3d, cfg 4): SQ pass"); table("/tmp/pc2/c_counter_collection.csv", "convbf1_kernel")
PY
# kNN: kernel stats + HBM traffic of one search at 100 k and 2 M rows: the screened regime, then the streaming regime
(cd $R && bash tools/profile_knn.sh) > $O/knn_profile.log 2>&1
K=$R/gpurun_out/prof_knn
for n in 100000 2000000; do
  python3 $R/tools/summarize_knn_pmc.py $K/knn_${n}_pmc_FETCH_SIZE.csv $K/knn_${n}_pmc_WRITE_SIZE.csv $n $P/pmc_knn_$n.json > $O/knn_pmc_$n.txt 2>&1
  cp $K/knn_${n}_kernels.txt $O/knn_${n}_kernels.txt
  export N=$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$n -o k -- python3 $R/tools/pmc_knn_stream.py > $O/knn_stream_$n.log 2>&1
  python3 $R/tools/summarize_trace.py /tmp/ks$n/k_kernel_trace.csv > $O/knn_stream_${n}_kernels.txt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ksf$n -o f -- python3 $R/tools/pmc_knn_stream.py > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ksw$n -o w -- python3 $R/tools/pmc_knn_stream.py > /dev/null 2>&1
  python3 $R/tools/summarize_knn_stream_pmc.py /tmp/ksf$n/f_counter_collection.csv /tmp/ksw$n/w_counter_collection.csv $n $P/pmc_knn_stream_$n.json > $O/knn_stream_pmc_$n.txt 2>&1
done
$R/tools/micro/l2_ingest > $O/l2_ingest.txt 2>&1
for c in 2 1 4 5; do python3 - <<PY
import json
d = json.loads([l for l in open("$O/bench_cfg$c.json") if l.startswith("{")][-1])
print("cfg $c:", d["ms_per_step"], "ms/step", d["rtf"], "x RT; host_io", d["host_io"]["ms_per_step"], "; knn", (d.get("roofline_knn") or {}).get("avg_search_ms"), "; knn stream frac", (d.get("roofline_knn_stream") or {}).get("frac"), "; dec", (d.get("decoder") or {}).get("ms"), "; cpu", (d.get("cpu_baseline") or {}).get("value"), "; roofline", (d.get("roofline") or {}).get("frac"))
PY
done
python3 -c "
import json
for f in ('bench_cfg2_again', 'bench_cfg4_again', 'bench_cfg2_inflight1', 'bench_cfg2_inflight2', 'bench_cfg2_100s'):
    d=json.loads([l for l in open('$O/'+f+'.json') if l.startswith('{')][-1]); print(f, d['ms_per_step'], 'ms/step; host_io', d['host_io']['ms_per_step'])"
tail -3 $O/pipeline_kernels.txt
