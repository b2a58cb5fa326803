#!/bin/bash
# ablation of the Winograd conv (k = 11, C = 128): which part of the loop costs what
export BENCH_C=${BENCH_C:-128} BENCH_K=11 RVC_WINO_R4=0   # the ablation builds exist for the F(4,3) form
for d in ${DBGS:-0 16 32 64 15 31 47 63 127}; do echo "DBG=$d"; RVC_WINO_DBG=$d python tools/bench_conv.py 2>&1 | grep "C="; done
