# round 4, call 51: the launch rule left to itself past its switch points
source tools/gpu_step.sh
step 600 gpurun_out/r4_51_tests.log python3 -m pytest tests/test_gpu_wave.py -x -q -m gpu -k "left_to_itself"
tail -6 gpurun_out/r4_51_tests.log
