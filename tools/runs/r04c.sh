mkdir -p gpurun_out/r04c
export MPLBACKEND=Agg
(time python -m pytest tests/test_parity_fullsize.py -m gpu -q -x -s -k "c3_fista_300 or c5_all_16 or c3_fista_30" > gpurun_out/r04c/gputests.log 2>&1); grep -v "^$" gpurun_out/r04c/gputests.log | tail -15
