# round 4, call 7: group race regression, optimizer tests against the independent traces, in-kernel stamp evidence for
# the bench default's launch time (first / last stamp build), the launch floor of an empty kernel in the same kind of graph
source tools/gpu_step.sh
step 600 gpurun_out/r4_07_tests.log python3 -m pytest tests/test_gpu_group.py tests/test_optimizer.py tests/test_gpu_fuzz.py -x -q -m gpu -k "not fp32 and not matches_the_oracle"
tail -3 gpurun_out/r4_07_tests.log
export GTOP_HIP_LIB=$PWD/build_var/libgtop_stamps2.so
step 200 gpurun_out/r4_07_stamps_span.txt python3 tools/stamps_span.py 1024
unset GTOP_HIP_LIB
cat gpurun_out/r4_07_stamps_span.txt
step 100 gpurun_out/r4_07_launch_floor.txt tools/ubench/launch_floor2
cat gpurun_out/r4_07_launch_floor.txt
