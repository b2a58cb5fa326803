#!/bin/bash
# builds tools/micro/mfma_cohab (victim W = the product library's wino kernel) and mfma_cohab_ablate (the ablation build's,
# whose RVC_WINO_FIX variants bisect the kernel: 64 no LDS-DMA, 128 __syncthreads, 256 s_nop behind the matrix instructions)
set -e
cd "$(dirname "$0")/../.."
L=codename-rvc-fork-3_amd/rvc_amd/_lib
F="--offload-arch=gfx950 -O3 -std=c++17 -I include -I codename-rvc-fork-3_amd/csrc tools/micro/mfma_cohab.hip -L $L -Wl,-rpath,\$ORIGIN/../../$L"
/opt/rocm/bin/hipcc $F -lrvc_amd -o tools/micro/mfma_cohab
/opt/rocm/bin/hipcc $F -l:librvc_amd_ablate.so -o tools/micro/mfma_cohab_ablate
