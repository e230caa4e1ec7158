# round 4, call 16: randomized launch-form / placement test of the optimizer loop, then a long fuzz hunt with fresh
# seeds on the final build (GTOP_FUZZ_EXTRA further seeds per randomised test, starting at GTOP_FUZZ_BASE)
source tools/gpu_step.sh
step 600 gpurun_out/r4_16_opt.log python3 -m pytest tests/test_optimizer.py -x -q -m gpu
tail -3 gpurun_out/r4_16_opt.log
export GTOP_FUZZ_EXTRA=1500 GTOP_FUZZ_BASE=700000
step 1100 gpurun_out/r4_16_fuzz.log python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -x
tail -5 gpurun_out/r4_16_fuzz.log
