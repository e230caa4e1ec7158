# round 4, call 4: the record builder with row reuse + both precisions in one pass
source tools/gpu_step.sh
step 300 gpurun_out/r4_04_tests.log python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_edt.py -x -q -m gpu
tail -2 gpurun_out/r4_04_tests.log
rm -rf gpurun_out/esdfprof
step 300 gpurun_out/r4_04_esdf_prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdfprof -- python3 tools/esdf_time.py 200 400
grep "per build" gpurun_out/r4_04_esdf_prof.log
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob("gpurun_out/esdfprof/*/*kernel_stats.csv"), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if "records" in r["Name"] or "esdf" in r["Name"]:
        print(r["Name"][:75], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
