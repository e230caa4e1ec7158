# round 4, call 38: one lane per segment (samples per lane 30, 64 / m trajectories per wavefront): parity tests, then its
# launch times against the current bodies on the large-batch workloads
source tools/gpu_step.sh
step 900 gpurun_out/r4_38_tests.log python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_api.py -x -q -m gpu
tail -8 gpurun_out/r4_38_tests.log
for spl in 0 30; do
  echo "=== spl $spl"
  GTOP_SPL=$spl timeout -k 10 300 python3 tools/variant_times_short.py 4096,6,f64 8192,6,f64 16384,6,f64 65536,6,f64 16384,6,f32 65536,6,f32 8192,12,f64 2>&1 | grep "B="
done > gpurun_out/r4_38_times.txt 2>&1
cat gpurun_out/r4_38_times.txt
