# round 4, call 34: optimizer fuzz with the two-per-wavefront loop in half the draws (up to 6 segments), 2 000 further seeds;
# then the whole fuzz file with 800 further seeds from a new base
source tools/gpu_step.sh
export GTOP_FUZZ_EXTRA=2000 GTOP_FUZZ_BASE=900000
step 1000 gpurun_out/r4_34_opt_fuzz.log python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k optimizer
tail -4 gpurun_out/r4_34_opt_fuzz.log
export GTOP_FUZZ_EXTRA=800 GTOP_FUZZ_BASE=1200000
step 1000 gpurun_out/r4_34_fuzz.log python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -x
tail -4 gpurun_out/r4_34_fuzz.log
