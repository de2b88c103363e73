mkdir -p gpurun_out/r04a
export MPLBACKEND=Agg
{
echo "== C1"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 1 5 20
echo "== C4"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 64 20 3
echo "== C4 shard 8"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 8 20 5
echo "== 760x1014 gray"; tools/gpu_ab_versions.sh _ab_r03 admm 1 760 1014 1 1 5 20
echo "== 1520x2028x3"; tools/gpu_ab_versions.sh _ab_r03 admm 1 1520 2028 3 1 50 2
} > gpurun_out/r04a/ab_pairing.log 2>&1
tail -40 gpurun_out/r04a/ab_pairing.log
(time python -m pytest tests/test_norm_scale.py tests/test_jit_hygiene.py tests/test_parity_small.py tests/test_abi_and_layout.py "tests/test_parity_large.py::test_no_compiler_on_the_gpu" "tests/test_parity_large.py::test_modules_unload_on_the_gpu" -m gpu -q -x > gpurun_out/r04a/gputests.log 2>&1); tail -15 gpurun_out/r04a/gputests.log
