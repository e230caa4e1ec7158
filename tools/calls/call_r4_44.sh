# round 4, call 44: three lanes / one lane per segment for SHORT trajectories (2 .. 5 segments: five lanes per segment fill
# only 20 .. 50 lanes of a wavefront with two of them) and the switch point for 7 .. 10 segments (-DGTOP_SPL10 build)
source tools/gpu_step.sh
export GTOP_HIP_LIB=$PWD/build_var/libgtop_spl10.so
for spl in 0 10 30; do
  echo "=== spl $spl"
  GTOP_SPL=$spl timeout -k 10 400 python3 tools/variant_times_short.py 16384,2,f64 16384,3,f64 16384,4,f64 16384,5,f64 16384,3,f32 16384,4,f32 16384,5,f32 4096,4,f64 4096,7,f64 4096,10,f64 4096,7,f32 3072,8,f64 2>&1 | grep "B="
done > gpurun_out/r4_44_times.txt 2>&1
cat gpurun_out/r4_44_times.txt
