# round 4, call 6: the windowed map update — bit-exactness against the oracle, and its time against the whole-map rebuild
source tools/gpu_step.sh
step 400 gpurun_out/r4_06_tests.log python3 -m pytest tests/test_window_update.py tests/test_optimizer.py tests/test_capi.py -x -q -m gpu
tail -3 gpurun_out/r4_06_tests.log
step 300 gpurun_out/r4_06_window_time.log python3 tools/window_time.py 200 400
cat gpurun_out/r4_06_window_time.log
