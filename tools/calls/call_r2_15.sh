source tools/gpu_step.sh
for lib in default W0; do
  if [ $lib = default ]; then unset GTOP_HIP_LIB; else export GTOP_HIP_LIB=$PWD/build_var/lib$lib.so; fi
  rm -rf gpurun_out/ab_$lib
  step 300 gpurun_out/ab_$lib.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$lib -- python3 bench.py --no-extras --no-cpu-baseline --steps 1000
  echo "$lib profiled bench line: $(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/ab_$lib.log)"
  step 100 gpurun_out/ab_plain_$lib.log python3 bench.py --no-extras --no-cpu-baseline --steps 1000
  echo "$lib plain bench line: $(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/ab_plain_$lib.log)"
done
unset GTOP_HIP_LIB
rm -rf gpurun_out/ab_floor
step 100 gpurun_out/ab_floor.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_floor -- tools/ubench/launch_floor2
python - <<'PY'
import csv,glob,os,numpy as np
for t in ("default","W0","floor"):
    f=sorted(glob.glob(f"gpurun_out/ab_{t}/**/*kernel_trace.csv",recursive=True), key=os.path.getmtime)[-1]
    rows=[r for r in csv.DictReader(open(f)) if ("gtop_eval" in r["Kernel_Name"] or "k_empty" in r["Kernel_Name"]) and r["Grid_Size_X"]=="65536"]
    st=np.array([int(r["Start_Timestamp"]) for r in rows]); en=np.array([int(r["End_Timestamp"]) for r in rows])
    o=np.argsort(st); st=st[o]; en=en[o]; d=(en-st)/1e3; sp=(st[1:]-st[:-1])/1e3
    print(t, len(rows), "duration mean %.2f median %.2f p10 %.2f p90 %.2f | spacing median %.2f"%(d.mean(),np.median(d),np.percentile(d,10),np.percentile(d,90),np.median(sp)))
PY
