mkdir -p gpurun_out/r04b
export MPLBACKEND=Agg
{
echo "== C1"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 1 5 20
echo "== C4"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 64 20 3
echo "== C4 shard 8"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 8 20 5
echo "== 760x1014 gray"; tools/gpu_ab_versions.sh _ab_r03 admm 1 760 1014 1 1 5 20
echo "== C2"; tools/gpu_ab_versions.sh _ab_r03 admm 1 3040 4056 3 1 40 1
} > gpurun_out/r04b/ab_pairing.log 2>&1
cat gpurun_out/r04b/ab_pairing.log | cut -c1-200
(time python -m pytest tests/test_norm_scale.py tests/test_jit_hygiene.py "tests/test_parity_large.py::test_no_compiler_on_the_gpu" "tests/test_parity_large.py::test_modules_unload_on_the_gpu" -m gpu -q -x > gpurun_out/r04b/gputests.log 2>&1); tail -5 gpurun_out/r04b/gputests.log
