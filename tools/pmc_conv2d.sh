#!/bin/bash
# matrix-pipe occupancy and wait breakdown of conv2d_mfma_kernel on selected U-Net shapes (SHAPES = indices into bench_conv2d.py)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export NO_TORCH=1 SHAPES=${SHAPES:-2,3,5}
rm -rf /tmp/pg
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d /tmp/pg -o g -- python3 $R/tools/bench_conv2d.py > /tmp/pg.log 2>&1
python3 - <<'PY'
import csv, collections
agg = collections.OrderedDict(); dur = {}
for r in csv.DictReader(open("/tmp/pg/g_counter_collection.csv")):
    if "conv2d_mfma" in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0][-28:], r.get("Grid_Size", r.get("Grid_Size_X")))
        agg.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    av = lambda n: sum(d[n]) / max(len(d[n]), 1)
    t = sum(dur[k]) / len(dur[k]); g = av("GRBM_GUI_ACTIVE"); wc = av("SQ_WAVE_CYCLES")
    print(f"{k[0]} grid {k[1]:>7s} {t/1e3:6.1f} us clk {g/8/t:.2f} GHz mfma_busy {av('SQ_VALU_MFMA_BUSY_CYCLES')/128/g:.3f} waves {av('SQ_WAVES'):.0f} | of wave-cycles: wait_inst {av('SQ_WAIT_INST_ANY')/wc:.3f} wait_lds {av('SQ_WAIT_INST_LDS')/wc:.3f} valu {av('SQ_ACTIVE_INST_VALU')/wc:.3f} lds {av('SQ_ACTIVE_INST_LDS')/wc:.3f}")
PY
