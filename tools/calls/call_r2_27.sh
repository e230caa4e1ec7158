source tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() {  # tag, bench args...
  local tag=$1; shift
  rm -rf gpurun_out/kprof_$tag
  step 300 gpurun_out/kprof_$tag.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$tag -- python3 bench.py --no-extras --no-cpu-baseline "$@"
  grep -o '"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' gpurun_out/kprof_$tag.log | tr '\n' ' '; echo
}
prof B16384_f32 --batch 16384 --dtype f32 --steps 500
prof B16384_f64 --batch 16384 --dtype f64 --steps 500
prof B8192_m12_g400_f64 --batch 8192 --segments 12 --grid 400 --density 0.04 --steps 300
bash tools/pmc_collect.sh B16384_f32 --batch 16384 --dtype f32
bash tools/pmc_collect.sh B16384_f64 --batch 16384 --dtype f64
bash tools/pmc_collect.sh B8192_m12_g400_f64 --batch 8192 --segments 12 --grid 400 --density 0.04
