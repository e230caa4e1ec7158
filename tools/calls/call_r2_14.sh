source tools/gpu_step.sh
prof() {  # tag, bench args...
  local tag=$1; shift
  rm -rf gpurun_out/kprof_$tag
  step 300 gpurun_out/kprof_$tag.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$tag -- python3 bench.py --no-extras --no-cpu-baseline "$@"
  grep -o '"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' gpurun_out/kprof_$tag.log | tr '\n' ' '; echo
}
prof default_B1024_f64 --steps 2000
prof B16384_f32 --batch 16384 --dtype f32 --steps 500
prof B16384_f64 --batch 16384 --dtype f64 --steps 500
prof B8192_m12_g400_f64 --batch 8192 --segments 12 --grid 400 --density 0.04 --steps 300
step 200 gpurun_out/stamps_wave_1024.txt env GTOP_HIP_LIB=$PWD/build_var/libS.so python tools/stamps_wave.py 1024
bash tools/pmc_collect.sh B1024_f64
bash tools/pmc_collect.sh B16384_f32 --batch 16384 --dtype f32
bash tools/pmc_collect.sh B16384_f64 --batch 16384 --dtype f64
bash tools/pmc_collect.sh B8192_m12_g400_f64 --batch 8192 --segments 12 --grid 400 --density 0.04
