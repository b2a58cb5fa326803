# rocprofv3 passes over tools/bench_knn.py (run on the GPU box through gpurun): kernel stats, then FETCH_SIZE / WRITE_SIZE in
# their own passes (the guide's rule: counters never together with other trace domains).  Output: gpurun_out/prof_knn/
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/prof_knn; mkdir -p $O
for N in 100000 2000000; do
  export N QS=1599
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk$N -o k -- python3 $R/tools/bench_knn.py > $O/bench_knn_$N.log 2>&1
  cp /tmp/pk$N/k_kernel_stats.csv $O/knn_${N}_kernel_stats.csv
  python3 $R/tools/summarize_trace.py /tmp/pk$N/k_kernel_trace.csv > $O/knn_${N}_kernels.txt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf$N -o f -- python3 $R/tools/bench_knn.py > /dev/null 2>&1
  grep -E "Kernel_Name|knn_" /tmp/pf$N/f_counter_collection.csv | cut -d, -f1-40 > $O/knn_${N}_pmc_FETCH_SIZE.csv
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw$N -o w -- python3 $R/tools/bench_knn.py > /dev/null 2>&1
  grep -E "Kernel_Name|knn_" /tmp/pw$N/w_counter_collection.csv | cut -d, -f1-40 > $O/knn_${N}_pmc_WRITE_SIZE.csv
done
cat $O/knn_100000_kernels.txt $O/knn_2000000_kernels.txt
head -3 $O/knn_100000_pmc_FETCH_SIZE.csv
