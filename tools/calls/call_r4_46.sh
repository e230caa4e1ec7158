# round 4, call 46: one lane per segment for trajectories of MORE than 12 segments (up to 64: a wavefront's slots), where the
# chunked body walks 12 segments at a time at five lanes per segment and its last chunk is mostly idle
source tools/gpu_step.sh
for spl in 0 30; do
  echo "=== spl $spl"
  GTOP_SPL=$spl timeout -k 10 400 python3 tools/variant_times_short.py 8192,13,f64 8192,17,f64 8192,24,f64 8192,25,f64 8192,32,f64 4096,40,f64 1024,13,f64 8192,13,f32 8192,17,f32 8192,24,f32 8192,32,f32 2>&1 | grep "B="
done > gpurun_out/r4_46_times.txt 2>&1
cat gpurun_out/r4_46_times.txt
