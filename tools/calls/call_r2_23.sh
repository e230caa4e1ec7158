source tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step 1100 gpurun_out/full_gpu_tests.log python -m pytest tests/ -x -q -m gpu || { tail -30 gpurun_out/full_gpu_tests.log; exit 1; }
tail -3 gpurun_out/full_gpu_tests.log
prof() {  # tag, bench args...
  local tag=$1; shift
  rm -rf gpurun_out/kprof_$tag
  step 300 gpurun_out/kprof_$tag.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$tag -- python3 bench.py --no-extras --no-cpu-baseline "$@"
  grep -o '"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' gpurun_out/kprof_$tag.log | tr '\n' ' '; echo
}
prof default_B1024_f64 --steps 2000
bash tools/pmc_collect.sh B1024_f64
step 300 gpurun_out/bench_default.json python bench.py --steps 20 --warmup 5
tail -1 gpurun_out/bench_default.json | cut -c1-600
