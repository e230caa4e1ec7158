# round 4, call 17: experiment — what would the segment count as a compile-time constant save (index arithmetic of the
# prologue)?  A/B on one box: shipped library against a -DGTOP_FIXED_M=6 build; then the rendezvous GPU tests after the
# in_call guard
source tools/gpu_step.sh
for rep in 1 2; do for L in grad_traj_optimization_amd/libgtop_hip.so build_var/libgtop_fixedm6.so; do
  echo "=== $L"; GTOP_HIP_LIB=$(realpath $L) timeout -k 10 300 python3 tools/variant_times_short.py 1024,6,f64 4096,6,f64 16384,6,f64 16384,6,f32 2>&1 | grep "B="
done; done > gpurun_out/r4_17_fixedm.txt 2>&1
cat gpurun_out/r4_17_fixedm.txt
step 600 gpurun_out/r4_17_tests.log python3 -m pytest tests/test_rendezvous.py tests/test_cpp_shim.py -x -q -m gpu
tail -3 gpurun_out/r4_17_tests.log
