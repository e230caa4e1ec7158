# round 3: x sweep variants of the ESDF build, per-kernel times from rocprofv3 (one run per library, same box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in grad_traj_optimization_amd/libgtop_hip.so $(ls build_var/libgtop_*.so) grad_traj_optimization_amd/libgtop_hip.so; do
  n=$(basename $L .so)
  rm -rf gpurun_out/esdfx_$n
  export GTOP_HIP_LIB=$(realpath $L)
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdfx_$n -- python3 tools/esdf_time.py 200 400 > gpurun_out/esdfx_$n.log 2>&1 || { echo "$n failed"; tail -3 gpurun_out/esdfx_$n.log; }
  f=$(ls -t gpurun_out/esdfx_$n/*/*kernel_stats.csv | head -1)
  echo "=== $n $(grep checksum gpurun_out/esdfx_$n.log | sed 's/.*checksum/checksum/' | tr '\n' ' ')"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'esdf_x' in n or 'esdf_y' in n or 'esdf_z' in n:
        print("   %-28s calls %3s min %7.1f max %7.1f avg %7.1f us" % (n.split('::')[-1][:28], r['Calls'], float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, float(r['AverageNs'])/1e3))
PY
done
