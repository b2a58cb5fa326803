#!/bin/bash
# HBM traffic of the dominant kernel (bench.py's `roofline.traffic`): FETCH_SIZE and WRITE_SIZE in separate passes
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r03pmc; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o f -- python3 $R/tools/pmc_conv.py > $O/pmc_conv_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o w -- python3 $R/tools/pmc_conv.py > /dev/null 2>&1
python3 $R/tools/summarize_pmc.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv > $O/pmc_wino.txt 2>&1
cat $O/pmc_wino.txt
grep -E "wino_conv_kernel<11, 2, 2, 8, 0|winobf_conv_kernel<11|conv_mfma_kernel<1," /tmp/pf/f_counter_collection.csv | head -60 > $O/pmc_FETCH_SIZE.csv
grep -E "wino_conv_kernel<11, 2, 2, 8, 0|winobf_conv_kernel<11|conv_mfma_kernel<1," /tmp/pw/w_counter_collection.csv | head -60 > $O/pmc_WRITE_SIZE.csv
