#!/bin/bash
# round 4, GPU call 3: cohabitation -- register-footprint variants + where the wrong words are; winobf2 ablations
mkdir -p gpurun_out/r04
timeout 900 tools/micro/mfma_cohab 100 > gpurun_out/r04/cohab_micro3.txt 2>&1
bash tools/ablate_winobf2.sh > gpurun_out/r04/winobf2_ablation.txt 2>&1
cat gpurun_out/r04/winobf2_ablation.txt | cut -c1-140
