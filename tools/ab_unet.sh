#!/bin/bash
# A/B: RMVPE U-Net convs in librvc_amd (conv2d.hip) vs MIOpen through PyTorch
mkdir -p gpurun_out/ab
for w in ${MODES:-0 1}; do
  RVC_NATIVE_UNET=$w python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-rooflines > gpurun_out/ab/unet$w.json 2> gpurun_out/ab/unet$w.err
  echo "NATIVE_UNET=$w rc=$?"; tail -2 gpurun_out/ab/unet$w.err
  RVC_NATIVE_UNET=$w python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-rooflines --inflight 1 > gpurun_out/ab/unet${w}_seq.json 2>/dev/null
done
