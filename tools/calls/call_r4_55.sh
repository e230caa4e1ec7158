# round 4, call 55: window update with the compact path's reset + marking fused (no gather kernel, no distance reset): bit-exact
# tests, then the timings
source tools/gpu_step.sh
step 600 gpurun_out/r4_55_tests.log python3 -m pytest tests/test_window_update.py tests/test_gpu_group.py tests/test_gpu_api.py -x -q -m gpu
tail -3 gpurun_out/r4_55_tests.log
step 300 gpurun_out/r4_55_window.log python3 tools/window_time.py 200 400
grep grid gpurun_out/r4_55_window.log
