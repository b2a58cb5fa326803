R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R; python -m pytest tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "frames or presplit or hubert" > $O/k12_tests.txt 2>&1; tail -2 $O/k12_tests.txt
python3 tools/bench_gemmbf.py 2>&1 | grep "conv layer\|ff1" | cut -c1-330
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/hf -o f -- python3 $R/tools/pmc_hubert_front.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/hw -o w -- python3 $R/tools/pmc_hubert_front.py > /dev/null 2>&1
python3 $R/tools/summarize_pmc_front.py /tmp/hf/f_counter_collection.csv /tmp/hw/w_counter_collection.csv | grep "^layer" | cut -c1-260
