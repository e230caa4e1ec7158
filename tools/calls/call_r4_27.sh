# round 4, call 27: the collective path after the bracket warm-up and the untimed rehearsal region: tests, then the numbers
source tools/gpu_step.sh
step 900 gpurun_out/r4_27_tests.log python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_api.py -x -q -m gpu
tail -4 gpurun_out/r4_27_tests.log
bash tools/calls/call_r4_24.sh
