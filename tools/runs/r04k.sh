mkdir -p gpurun_out/r04k; export TMPDIR=/tmp; R=$PWD
cd /tmp
rocprofv3 --kernel-trace -f csv -d $R/gpurun_out/r04k/c1 -o t -- python $R/tools/probe/c1_trace.py 270 480 3 5 > $R/gpurun_out/r04k/c1.log 2>&1
cd $R
tail -1 gpurun_out/r04k/c1.log
f=$(find gpurun_out/r04k/c1 -name "*kernel_trace.csv" | head -1); python tools/probe/trace_gaps.py $f 440 | tee gpurun_out/r04k/c1_gaps.txt
python tools/probe/c1_trace.py 270 480 3 5 | tail -1
find gpurun_out/r04k -name "*.csv" -size +2M -delete
