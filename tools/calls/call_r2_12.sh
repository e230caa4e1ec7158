source tools/gpu_step.sh
step 600 gpurun_out/pytest12.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_edt.py tests/test_golden.py -q -m gpu
python -c "import __graft_entry__ as g; g.smoke()"
tail -4 gpurun_out/pytest12.log
for lib in default; do
  if [ $lib = default ]; then unset GTOP_HIP_LIB; else export GTOP_HIP_LIB=$PWD/build_var/lib$lib.so; fi
  rm -rf gpurun_out/esdf_prof_$lib
  step 300 gpurun_out/esdf_$lib.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdf_prof_$lib -- python3 tools/esdf_time.py
  grep "grid" gpurun_out/esdf_$lib.log
done
python - <<'PY'
import csv,glob
for lib in ("default",):
    f=glob.glob(f"gpurun_out/esdf_prof_{lib}/**/*kernel_stats.csv",recursive=True)[0]
    print("==",lib)
    for r in csv.DictReader(open(f)):
        if "esdf" in r["Name"]:
            print(f'  {r["Name"][:60]:60s} calls {r["Calls"]:>3s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
PY
