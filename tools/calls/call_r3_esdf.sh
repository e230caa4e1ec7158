# round 3: the ESDF build with the packed 16-bit y sweep against the previous library (same box), exactness tests, kernel times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_golden.py -q -x -k "esdf or sdf or golden or full_size or configs4 or update" > gpurun_out/r3_esdf_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r3_esdf_tests.log
for L in build_var/libgtop_prev.so grad_traj_optimization_amd/libgtop_hip.so build_var/libgtop_prev.so grad_traj_optimization_amd/libgtop_hip.so; do
  echo "=== $L"; GTOP_HIP_LIB=$(realpath $L) python tools/esdf_time.py 200 400 2>&1 | grep "per build\|checksum"
done
rm -rf gpurun_out/esdfprof
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdfprof -- python3 tools/esdf_time.py 200 400 > gpurun_out/esdf_prof.log 2>&1
f=$(ls -t gpurun_out/esdfprof/*/*kernel_stats.csv | head -1); cut -d, -f1-4 $f | cut -c1-150
