ROWS=${ROWS:-"1024,6,f64 16384,6,f64 1024,13,f64 16384,6,f64,dyn"}
LIBS=${LIBS:-"base NO_GQ_PIN NO_TINY_OPAQUE NEW"}
for i in 1 2; do for L in $LIBS; do
  P=build_var/libgtop_$L.so; [ $L = NEW ] && P=grad_traj_optimization_amd/libgtop_hip.so
  echo "=== $L"; GTOP_HIP_LIB=$(realpath $P) python tools/variant_times_short.py $ROWS 2>&1 | grep "B="
done; done
