source tools/gpu_step.sh
step 600 gpurun_out/pytest11.log python -m pytest tests/test_gpu_parity.py::test_esdf_bit_exact tests/test_gpu_api.py -q -m gpu -k "esdf or sdf or 400"
tail -4 gpurun_out/pytest11.log
for lib in default EY1X1 EY0X0; do
  if [ $lib = default ]; then unset GTOP_HIP_LIB; else export GTOP_HIP_LIB=$PWD/build_var/lib$lib.so; fi
  step 300 gpurun_out/esdf_$lib.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdf_prof_$lib -- python3 tools/esdf_time.py
  grep "grid" gpurun_out/esdf_$lib.log
  f=$(find gpurun_out/esdf_prof_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; grep -i "esdf" $f | cut -d, -f1-4,7,8 | head -8
done
