# round 4, call 50: experiment — two lanes per segment (samples per lane 15, 32 / m trajectories per wavefront, fp64 only) in a
# -DGTOP_SPL15 build: parity, then launch times against the rule's bodies for 6 segments (five trajectories per wavefront)
source tools/gpu_step.sh
export GTOP_HIP_LIB=$PWD/build_var/libgtop_spl15.so
CHECK_SPL=15 timeout -k 10 300 python3 tools/proto/spl10_check.py 2>&1 | grep float64 | awk '{print $NF}' | sort | uniq -c
for spl in 0 15 0 15; do
  echo "=== spl $spl"
  GTOP_SPL=$spl timeout -k 10 400 python3 tools/variant_times_short.py 8192,6,f64 16384,6,f64 65536,6,f64 16384,4,f64 16384,5,f64 8192,8,f64 8192,11,f64 8192,16,f64 2>&1 | grep "B="
done > gpurun_out/r4_50_times.txt 2>&1
cat gpurun_out/r4_50_times.txt
