#!/bin/bash
tools/gpu_evidence.sh r05end > gpurun_out/r05end_console.log 2>&1
tail -40 gpurun_out/r05end_console.log
