#!/bin/bash
tools/gpu_evidence.sh r05fin > gpurun_out/r05fin_console.log 2>&1
(time python -m pytest tests/test_parity_fullsize_long.py -m gpu_long -q -s > gpurun_out/r05fin/gpu_long.log 2>&1); tail -5 gpurun_out/r05fin/gpu_long.log
tail -40 gpurun_out/r05fin_console.log
