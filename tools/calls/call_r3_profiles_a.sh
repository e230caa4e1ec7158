# round 3: rocprofv3 kernel stats of the bench workloads + PMC passes for the headline and the fp32 roofline run
source tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() {  # tag, bench args...
  local tag=$1; shift
  rm -rf gpurun_out/kprof_$tag
  step 300 gpurun_out/kprof_$tag.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$tag -- python3 bench.py --no-extras --no-cpu-baseline "$@"
  grep -o '"probe": [0-9.]*\|"frac": [0-9.]*' gpurun_out/kprof_$tag.log | head -3 | tr '\n' ' '; echo
}
prof default_B1024_f64 --steps 2000 --warmup 100
prof B16384_f32 --batch 16384 --dtype f32 --steps 500
prof B16384_f64 --batch 16384 --dtype f64 --steps 500
prof B8192_m12_g400_f64 --batch 8192 --segments 12 --grid 400 --density 0.04 --steps 300
bash tools/pmc_collect.sh B1024_f64
bash tools/pmc_collect.sh B16384_f32 --batch 16384 --dtype f32
