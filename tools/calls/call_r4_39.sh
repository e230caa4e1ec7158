# round 4, call 39: where (if anywhere) one lane per segment pays: fp32 and fp64 at larger batches, 6 and 12 segments
source tools/gpu_step.sh
for spl in 0 30 0 30; do
  echo "=== spl $spl"
  GTOP_SPL=$spl timeout -k 10 400 python3 tools/variant_times_short.py 32768,6,f32 65536,6,f32 131072,6,f32 8192,12,f32 32768,12,f32 131072,6,f64 32768,12,f64 2>&1 | grep "B="
done > gpurun_out/r4_39_times.txt 2>&1
cat gpurun_out/r4_39_times.txt
