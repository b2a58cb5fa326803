export RVC_AMD_LIB=codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so
for i in 1 2 3; do
  for tp in 0 1; do
    RVC_LBF_TP=$tp timeout 200 python3 bench.py --no-cpu-baseline --no-rooflines 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('RVC_LBF_TP=$tp ms_per_step', d['ms_per_step'])"
  done
done
