mkdir -p gpurun_out
for mode in 6 14 4 12 0 8; do for sl in 0 8; do tools/probe/row_pattern $mode $sl 20; done; done > gpurun_out/r05j2_rowpattern.log 2>&1
cat gpurun_out/r05j2_rowpattern.log
