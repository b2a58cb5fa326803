#!/bin/bash
# where does attention_bf_kernel's time go (ablation build; wrong results): kernel-trace average per variant
cd /tmp; export TMPDIR=/tmp
AB=$GRAFT_REPO_ROOT/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so
for d in 0 1 2 4 8 16 7; do
  rm -rf /tmp/pa$d
  RVC_AMD_LIB=$AB RVC_ATT_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa$d -o a -- python3 $GRAFT_REPO_ROOT/tools/bench_attention.py > /dev/null 2>&1
  echo "RVC_ATT_DBG=$d: $(grep attention_bf_kernel /tmp/pa$d/a_kernel_stats.csv | cut -d, -f2-4)"
done
