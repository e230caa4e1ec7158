# round 4, call 40: the whole GPU suite with the one-lane-per-segment body in the library, and a fuzz hunt that draws it
source tools/gpu_step.sh
step 1100 gpurun_out/r4_40_tests.log python3 -m pytest tests -x -q -m gpu
tail -4 gpurun_out/r4_40_tests.log
export GTOP_FUZZ_EXTRA=1200 GTOP_FUZZ_BASE=1500000
step 1000 gpurun_out/r4_40_fuzz.log python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k "oracle or fp32"
tail -4 gpurun_out/r4_40_fuzz.log
