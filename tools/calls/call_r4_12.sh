# round 4, call 12: the optimizer loop with two trajectories per wavefront: its tests, then wall times by batch
# (spl pinned 3 = one per wavefront, 6 = two per wavefront) to place the switch point
source tools/gpu_step.sh
step 900 gpurun_out/r4_12_tests.log python3 -m pytest tests/test_optimizer.py tests/test_gpu_kino.py tests/test_gpu_api.py -x -q -m gpu
tail -5 gpurun_out/r4_12_tests.log
step 400 gpurun_out/r4_12_opt_time.log python3 tools/opt_time.py 1024 2048 3072 4096 8192 16384
cat gpurun_out/r4_12_opt_time.log
