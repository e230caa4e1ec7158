# round 4, call 41: the whole GPU suite again (launch geometry value test updated; the 64-bit-index body of the one-lane geometry)
source tools/gpu_step.sh
step 1100 gpurun_out/r4_41_tests.log python3 -m pytest tests -x -q -m gpu
tail -4 gpurun_out/r4_41_tests.log
