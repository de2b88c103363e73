#!/bin/bash
tools/gpu_evidence.sh r05last > gpurun_out/r05last_console.log 2>&1
tail -40 gpurun_out/r05last_console.log
