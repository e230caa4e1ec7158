# round 4, call 45: three lanes per segment in the launch rule: the whole GPU suite, then launch times with the rule left to
# itself against the pinned geometries for the cases the switch points need (short trajectories at 8 192, fp32 at large B)
source tools/gpu_step.sh
step 1100 gpurun_out/r4_45_tests.log python3 -m pytest tests -x -q -m gpu
tail -5 gpurun_out/r4_45_tests.log
for spl in 6 10 30; do
  echo "=== spl $spl"
  GTOP_SPL=$spl timeout -k 10 400 python3 tools/variant_times_short.py 8192,3,f64 8192,4,f64 8192,5,f64 8192,4,f32 4096,4,f32 4096,3,f64 131072,3,f32 131072,5,f32 65536,4,f32 65536,8,f32 2>&1 | grep "B="
done > gpurun_out/r4_45_times.txt 2>&1
cat gpurun_out/r4_45_times.txt
