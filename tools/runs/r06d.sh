#!/bin/bash
# round 6, run d: the whole GPU suite on the pruned tree, batches as sub-batches on several streams, short bench
mkdir -p gpurun_out/r06d
(time python -m pytest tests -m gpu -q -s --durations=12 > gpurun_out/r06d/gputests.log 2>&1) 2>&1 | grep real
grep -E "passed|failed|FAILED|Error" gpurun_out/r06d/gputests.log | tail -8
grep -E "^[0-9.]+s (call|setup)" gpurun_out/r06d/gputests.log | head -12
python tools/probe/two_streams.py 64 20 3 > gpurun_out/r06d/streams64.log 2>&1; cut -c1-250 gpurun_out/r06d/streams64.log | grep -v amdgpu.ids
python tools/probe/two_streams.py 16 20 10 > gpurun_out/r06d/streams16.log 2>&1; cut -c1-250 gpurun_out/r06d/streams16.log | grep -v amdgpu.ids
(time python bench.py --steps 5 --warmup 2 > gpurun_out/r06d/bench.json 2> gpurun_out/r06d/bench.log) 2>&1 | grep real
tail -4 gpurun_out/r06d/bench.log
python tools/kernel_summary.py gpurun_out/r06d/bench.json; python tools/other_summary.py gpurun_out/r06d/bench.json
