# round 4, call 36: experiment — streaming (non-temporal) stores for the gradient: does the kernel boundary get cheaper?
source tools/gpu_step.sh
for rep in 1 2 3; do for L in grad_traj_optimization_amd/libgtop_hip.so build_var/libgtop_ntst.so; do
  echo "=== $L"; GTOP_HIP_LIB=$(realpath $L) timeout -k 10 300 python3 tools/variant_times_short.py 1024,6,f64 4096,6,f64 16384,6,f64 16384,6,f32 2>&1 | grep "B="
done; done > gpurun_out/r4_36_ntst.txt 2>&1
cat gpurun_out/r4_36_ntst.txt
