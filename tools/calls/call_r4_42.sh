# round 4, call 42: experiment — three lanes per segment (samples per lane 10, 21 / m trajectories per wavefront) in a
# -DGTOP_SPL10 build: parity against the oracle, then launch times against the launch rule's bodies
source tools/gpu_step.sh
export GTOP_HIP_LIB=$PWD/build_var/libgtop_spl10.so
step 300 gpurun_out/r4_42_check.log python3 tools/proto/spl10_check.py
grep -c OK gpurun_out/r4_42_check.log; grep BAD gpurun_out/r4_42_check.log | head
for spl in 0 10 0 10; do
  echo "=== spl $spl"
  GTOP_SPL=$spl timeout -k 10 400 python3 tools/variant_times_short.py 4096,6,f32 8192,6,f32 16384,6,f32 32768,6,f32 65536,6,f32 16384,6,f64 8192,10,f32 2>&1 | grep "B="
done > gpurun_out/r4_42_times.txt 2>&1
cat gpurun_out/r4_42_times.txt
