# round 4, call 49: fuzz hunt with the three-lane and one-lane bodies drawn in two fifths (fp64) / two thirds (fp32) of the runs
source tools/gpu_step.sh
export GTOP_FUZZ_EXTRA=1500 GTOP_FUZZ_BASE=2100000
step 1100 gpurun_out/r4_49_fuzz.log python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k "oracle or fp32"
tail -4 gpurun_out/r4_49_fuzz.log
