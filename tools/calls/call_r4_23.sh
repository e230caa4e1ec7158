# round 4, call 23: the all-gather as point-to-point stores (gtop_push_rows): unit test, the rank path at RCCL world size 1,
# two real processes on one card mapping each other's buffers
source tools/gpu_step.sh
step 600 gpurun_out/r4_23_unit.log python3 -m pytest tests/test_gpu_api.py -x -q -m gpu -k push_rows
tail -3 gpurun_out/r4_23_unit.log
step 900 gpurun_out/r4_23_tests.log python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu
tail -15 gpurun_out/r4_23_tests.log
