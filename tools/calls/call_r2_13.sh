source tools/gpu_step.sh
step 600 gpurun_out/pytest13.log python -m pytest tests/test_gpu_edt.py tests/test_setup_post.py tests/test_capi.py -q -m gpu
tail -6 gpurun_out/pytest13.log
rm -rf gpurun_out/f4_prof
step 300 gpurun_out/f4.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f4_prof -- python3 tools/f4_time.py
grep "edt_query\|trajectory_stats" gpurun_out/f4.log
python - <<'PY'
import csv,glob,os
f=sorted(glob.glob("gpurun_out/f4_prof/**/*kernel_stats.csv",recursive=True), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("edt_query","eval_trajectories","coefficients")):
        print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>3s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
