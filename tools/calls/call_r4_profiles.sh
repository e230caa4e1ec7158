# round 4: rocprofv3 kernel stats of the four bench workloads and of the driver's own command, PMC passes for all four
# (summarised on the box; the per-dispatch CSVs are too large to bring back), the variant list, map-build / query / window
# kernels, the in-kernel stamp span and the launch floor.  Outputs under gpurun_out/, the judged copies under profiles/r4/.
source tools/gpu_step.sh
prof() {  # tag, bench args...
  local tag=$1; shift
  rm -rf gpurun_out/kprof_$tag
  step 300 gpurun_out/kprof_$tag.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$tag -- python3 bench.py "$@"
  find gpurun_out/kprof_$tag -name '*kernel_trace.csv' -delete   # (tens of MB; the stats CSV is what is kept)
  f=$(ls -t gpurun_out/kprof_$tag/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_$tag.csv; head -2 $f | tail -1 | cut -d, -f2-8
  grep '^{' gpurun_out/kprof_$tag.log > gpurun_out/bench_under_rocprof_$tag.json
}
prof driver_cmd --gpus 1 --steps 20 --warmup 5
prof default_B1024_f64 --no-extras --no-cpu-baseline --steps 2000 --warmup 100
prof B16384_f32 --no-extras --no-cpu-baseline --batch 16384 --dtype f32 --steps 500
prof B16384_f64 --no-extras --no-cpu-baseline --batch 16384 --dtype f64 --steps 500
prof B8192_m12_g400_f64 --no-extras --no-cpu-baseline --batch 8192 --segments 12 --grid 400 --density 0.04 --steps 300
pmc() {  # tag, bench args...: the seven passes, then the summary
  local tag=$1; shift
  bash tools/pmc_collect.sh $tag "$@" > gpurun_out/pmc_$tag.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/summary_$tag.json
  rm -rf gpurun_out/pmc_$tag
  grep -c done gpurun_out/pmc_$tag.log
}
pmc B1024_f64
pmc B16384_f32 --batch 16384 --dtype f32
pmc B16384_f64 --batch 16384 --dtype f64
pmc B8192_m12_g400_f64 --batch 8192 --segments 12 --grid 400 --density 0.04
step 400 gpurun_out/r4_variant_times.txt python3 tools/variant_times.py
rm -rf gpurun_out/esdfprof gpurun_out/f4prof
step 300 gpurun_out/r4_esdf_prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdfprof -- python3 tools/esdf_time.py 200
f=$(ls -t gpurun_out/esdfprof/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_esdf_200.csv
rm -rf gpurun_out/esdfprof
step 300 gpurun_out/r4_esdf_prof400.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdfprof -- python3 tools/esdf_time.py 400
f=$(ls -t gpurun_out/esdfprof/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_esdf_400.csv
step 300 gpurun_out/r4_f4_prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f4prof -- python3 tools/f4_time.py
f=$(ls -t gpurun_out/f4prof/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_f4.csv
bash tools/calls/call_r3_edt_pmc.sh > gpurun_out/r4_edt_pmc.txt 2>&1
step 300 gpurun_out/r4_window_time.txt python3 tools/window_time.py 200 400
export GTOP_HIP_LIB=$PWD/build_var/libgtop_stamps2.so
step 200 gpurun_out/r4_stamps_span.txt python3 tools/stamps_span.py 1024
unset GTOP_HIP_LIB
step 100 gpurun_out/r4_launch_floor.txt tools/ubench/launch_floor2
step 300 gpurun_out/r4_bench_driver.log python3 bench.py --gpus 1 --steps 20 --warmup 5
grep '^{' gpurun_out/r4_bench_driver.log > gpurun_out/r4_bench_driver.json
