# round 4, call 15: window update captured into a hipGraph (scratch allocated once, for any window)
source tools/gpu_step.sh
step 600 gpurun_out/r4_15_tests.log python3 -m pytest tests/test_window_update.py tests/test_gpu_group.py tests/test_setup_post.py tests/test_gpu_edt.py -x -q -m gpu
tail -5 gpurun_out/r4_15_tests.log
