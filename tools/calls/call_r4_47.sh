# round 4, call 47: one lane per segment past 12 segments: the remaining lengths and the batch switch point
source tools/gpu_step.sh
for spl in 0 30; do
  echo "=== spl $spl"
  GTOP_SPL=$spl timeout -k 10 400 python3 tools/variant_times_short.py 8192,22,f64 8192,33,f64 8192,36,f64 4096,48,f64 4096,64,f64 2048,13,f64 4096,13,f64 2048,17,f64 4096,17,f64 4096,32,f64 2048,32,f64 4096,13,f32 8192,36,f32 2>&1 | grep "B="
done > gpurun_out/r4_47_times.txt 2>&1
cat gpurun_out/r4_47_times.txt
