#!/bin/bash
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
# effective clock + matrix-pipe occupancy of the Winograd conv and its ablations (RVC_WINO_DBG), C=128 K=11
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export BENCH_C=${BENCH_C:-128} BENCH_K=11
for d in ${DBGS:-0 15 127}; do
export RVC_WINO_DBG=$d
rm -rf /tmp/pg
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pg -o g -- python3 $R/tools/bench_conv.py > /dev/null 2>&1
echo "DBG=$d"
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/pg/g_counter_collection.csv")):
    if "wino_conv" in r["Kernel_Name"] or "conv_mfma_kernel" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    av = lambda n: sum(d[n]) / max(len(d[n]), 1)
    t = sum(dur[k]) / len(dur[k]); g = av("GRBM_GUI_ACTIVE"); wc = av("SQ_WAVE_CYCLES")
    print(f"{k} {t/1e3:7.1f} us clk {g/t:.3f} GHz mfma_busy {av('SQ_VALU_MFMA_BUSY_CYCLES')/1024/g:.3f} | of wave-cycles: wait_inst {av('SQ_WAIT_INST_ANY')/wc:.3f} valu {av('SQ_ACTIVE_INST_VALU')/wc:.3f} lds {av('SQ_ACTIVE_INST_LDS')/wc:.3f} vmem {av('SQ_ACTIVE_INST_VMEM')/wc:.3f} | valu insts {av('SQ_INSTS_VALU'):.3e} wave_cycles {wc:.3e}")
PY
done
