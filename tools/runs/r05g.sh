mkdir -p gpurun_out
tools/gpu_ab.sh fista 1 3040 4056 3 1 40 2 "stagger=0" "stagger=3" "stagger=6" "stagger=9" "stagger=12" "stagger=0" "stagger=6" "stagger=18" > gpurun_out/r05g_stagger.log 2>&1
grep "best" gpurun_out/r05g_stagger.log
