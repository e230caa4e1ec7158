# round 4, call 43: three lanes per segment (samples per lane 10) for trajectories of 7 .. 10 segments, where five lanes per
# segment leave 14 .. 29 lanes of a wavefront idle: launch times against the launch rule's bodies (-DGTOP_SPL10 build)
source tools/gpu_step.sh
export GTOP_HIP_LIB=$PWD/build_var/libgtop_spl10.so
for spl in 0 10 30; do
  echo "=== spl $spl"
  GTOP_SPL=$spl timeout -k 10 400 python3 tools/variant_times_short.py 2048,7,f64 8192,7,f64 2048,8,f64 8192,8,f64 8192,9,f64 2048,10,f64 8192,10,f64 8192,7,f32 8192,8,f32 2048,10,f32 8192,10,f32 1024,7,f64 1024,10,f64 2>&1 | grep "B="
done > gpurun_out/r4_43_times.txt 2>&1
cat gpurun_out/r4_43_times.txt
