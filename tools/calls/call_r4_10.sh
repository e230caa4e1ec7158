# round 4, call 10: record-builder variants (rows per lane, non-temporal stores) on one box: whole-map rebuild times;
# then the GPU suite with the new launch-rule switch points
source tools/gpu_step.sh
for rep in 1 2; do
for L in grad_traj_optimization_amd/libgtop_hip.so build_var/libgtop_recTY4.so build_var/libgtop_recTY16.so build_var/libgtop_recTY32.so build_var/libgtop_recNT.so; do
  echo "=== $L"; GTOP_HIP_LIB=$(realpath $L) python3 tools/esdf_time.py 200 400 2>&1 | grep "per build"
done; done > gpurun_out/r4_10_rec_variants.txt 2>&1
cat gpurun_out/r4_10_rec_variants.txt
step 1000 gpurun_out/r4_10_tests.log python3 -m pytest tests -x -q -m gpu
tail -4 gpurun_out/r4_10_tests.log
