R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r1b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python3 $R/bench.py --steps 5 --warmup 2 > $R/gpurun_out/r1b/bench_under_rocprof.log 2>&1
cp /tmp/pb/b_kernel_stats.csv $R/gpurun_out/r1b/bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o f -- python3 $R/tools/pmc_conv.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o w -- python3 $R/tools/pmc_conv.py > /dev/null 2>&1
ls /tmp/pf /tmp/pw
python3 $R/tools/summarize_pmc.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv > $R/gpurun_out/r1b/pmc_conv.txt 2>&1
cat $R/gpurun_out/r1b/pmc_conv.txt
grep -E "conv_mfma_kernel<(1|11)," /tmp/pf/f_counter_collection.csv | head -40 > $R/gpurun_out/r1b/pmc_FETCH_SIZE.csv
grep -E "conv_mfma_kernel<(1|11)," /tmp/pw/w_counter_collection.csv | head -40 > $R/gpurun_out/r1b/pmc_WRITE_SIZE.csv
head -1 /tmp/pf/f_counter_collection.csv > $R/gpurun_out/r1b/pmc_header.csv
rocprofv3 --kernel-trace --output-format csv -d /tmp/d1 -o t -- python3 $R/tools/profile_decoder.py >/dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/d1/t_kernel_trace.csv > $R/gpurun_out/r1b/decoder_kernels_nsf.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/d2 -o t -- python3 $R/tools/profile_decoder.py RefineGAN >/dev/null 2>&1; python3 $R/tools/summarize_trace.py /tmp/d2/t_kernel_trace.csv > $R/gpurun_out/r1b/decoder_kernels_refine.txt
python3 $R/bench.py --steps 8 --warmup 2 > $R/gpurun_out/r1b/bench.json 2>/dev/null
tail -1 $R/gpurun_out/r1b/bench.json | cut -c1-200
head -12 $R/gpurun_out/r1b/bench_kernel_stats.csv | cut -c1-160
