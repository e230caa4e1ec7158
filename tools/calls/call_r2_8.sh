source tools/gpu_step.sh
for lib in default REP; do
  if [ $lib = default ]; then unset GTOP_HIP_LIB; else export GTOP_HIP_LIB=$PWD/build_var/lib$lib.so; fi
  step 120 gpurun_out/rep_$lib.json python bench.py --no-extras --no-cpu-baseline --steps 1000
  grep -o '"avg_launch_us": [0-9.]*' gpurun_out/rep_$lib.json
done
