# round 4, call 58: the optimizer loop with 21 / m trajectories per wavefront (three lanes per segment): its tests
source tools/gpu_step.sh
step 900 gpurun_out/r4_58_tests.log python3 -m pytest tests/test_optimizer.py -x -q -m gpu
tail -12 gpurun_out/r4_58_tests.log
