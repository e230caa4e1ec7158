#!/bin/bash
# A/B of ESDF build variants: kernel times at 200^3 / 400^3 per library
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/grad_traj_optimization_amd/libgtop_$v.so
  export GTOP_HIP_LIB=$lib
  rm -rf gpurun_out/esdf_ab_$v
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdf_ab_$v -- python3 tools/esdf_time.py > gpurun_out/esdf_ab_$v.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/esdf_ab_$v.log; exit 1; }
  echo "== $v: $(grep -o 'checksum [0-9.]*' gpurun_out/esdf_ab_$v.log | tr '\n' ' ')"
  python3 - "$v" <<'PY'
import csv, glob, sys
v = sys.argv[1]
f = sorted(glob.glob(f"gpurun_out/esdf_ab_{v}/*/*kernel_stats.csv"))[-1]
tot = [0.0, 0.0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("esdf_x", "esdf_y", "esdf_z", "esdf_rows")):
        name = n.split("(anonymous namespace)::")[1].split("(")[0]
        lo, hi = float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3
        if "esdf_z_small_kernel<4>" in n: tot[0] += lo; print(f"    {name:28s} 200^3 {lo:7.1f} us"); continue
        if "esdf_z" in n: tot[1] += hi; print(f"    {name:28s} 400^3 {hi:7.1f} us"); continue
        tot[0] += lo; tot[1] += hi
        print(f"    {name:28s} 200^3 {lo:7.1f} us   400^3 {hi:7.1f} us")
print(f"    build total: 200^3 {tot[0]:.1f} us, 400^3 {tot[1]:.1f} us")
PY
done
