# round 4, call 3: corner records measured — the builder's own time (kernel stats of the map build), the f4 query kernel
# (time + fabric counters), the fabric counters of configs[4] and configs[1]
source tools/gpu_step.sh
rm -rf gpurun_out/esdfprof gpurun_out/f4prof
step 300 gpurun_out/r4_03_esdf_prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdfprof -- python3 tools/esdf_time.py 200 400
f=$(ls -t gpurun_out/esdfprof/*/*kernel_stats.csv | head -1); cut -d, -f1-4 $f | cut -c1-170 > gpurun_out/r4_03_esdf_kernels.txt; cat gpurun_out/r4_03_esdf_kernels.txt
grep "per build" gpurun_out/r4_03_esdf_prof.log
step 200 gpurun_out/r4_03_f4.log python3 tools/f4_time.py
cat gpurun_out/r4_03_f4.log
bash tools/calls/call_r3_edt_pmc.sh > gpurun_out/r4_03_edt_pmc.txt 2>&1; tail -3 gpurun_out/r4_03_edt_pmc.txt
pmc() {  # tag, bench args...: the seven passes, then the summary; the per-dispatch CSVs (tens of MB) stay on the box
  local tag=$1; shift
  bash tools/pmc_collect.sh $tag "$@" > gpurun_out/pmc_$tag.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/summary_$tag.json
  rm -rf gpurun_out/pmc_$tag
  grep -c done gpurun_out/pmc_$tag.log
}
pmc B8192_m12_g400_f64 --batch 8192 --segments 12 --grid 400 --density 0.04
pmc B1024_f64
