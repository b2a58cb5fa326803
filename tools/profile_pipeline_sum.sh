# one warm cfg-2 utterance under rocprofv3: the kernel list (tools/profile_pipeline.py + summarize_trace.py) -> gpurun_out/pp_kernels.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o t -- python3 $R/tools/profile_pipeline.py > $R/gpurun_out/pp.log 2>&1
python3 $R/tools/summarize_trace.py /tmp/pp/t_kernel_trace.csv 0 all lastgap > $R/gpurun_out/pp_kernels.txt
tail -1 $R/gpurun_out/pp_kernels.txt
