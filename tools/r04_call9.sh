#!/bin/bash
mkdir -p gpurun_out/r04
timeout 3300 python -m pytest tests/ -m gpu -q > gpurun_out/r04/gpu_suite_b.txt 2>&1
tail -8 gpurun_out/r04/gpu_suite_b.txt
