# round 4, call 28: push path with explicit peer-access enabling; the forced-failure fallback with two processes
source tools/gpu_step.sh
step 900 gpurun_out/r4_28_tests.log python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu
tail -6 gpurun_out/r4_28_tests.log
