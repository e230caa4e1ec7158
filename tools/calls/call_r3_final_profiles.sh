# round 3, final tree: rocprofv3 kernel stats of the four bench workloads and of the driver's own command, PMC passes for all four
source tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() {  # tag, bench args...
  local tag=$1; shift
  rm -rf gpurun_out/kprof_$tag
  step 300 gpurun_out/kprof_$tag.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$tag -- python3 bench.py "$@"
  find gpurun_out/kprof_$tag -name '*kernel_trace.csv' -delete   # (tens of MB; the stats CSV is what is kept)
  f=$(ls -t gpurun_out/kprof_$tag/*/*kernel_stats.csv | head -1); head -2 $f | tail -1 | cut -d, -f2-8
}
prof driver_cmd --gpus 1 --steps 20 --warmup 5
prof default_B1024_f64 --no-extras --no-cpu-baseline --steps 2000 --warmup 100
prof B16384_f32 --no-extras --no-cpu-baseline --batch 16384 --dtype f32 --steps 500
prof B16384_f64 --no-extras --no-cpu-baseline --batch 16384 --dtype f64 --steps 500
prof B8192_m12_g400_f64 --no-extras --no-cpu-baseline --batch 8192 --segments 12 --grid 400 --density 0.04 --steps 300
pmc() {  # tag, bench args...: the seven passes, then the summary; the per-dispatch CSVs (tens of MB) stay on the box
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
