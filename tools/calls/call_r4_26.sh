# round 4, call 26: is the short collective region's spread a first-region effect?  the same region six times per process
source tools/gpu_step.sh
for rep in 1 2 3; do
for mode in "GTOP_BENCH_GATHER=push" "GTOP_BENCH_GATHER=library"; do
  echo "=== $mode"
  env GTOP_BENCH_FORCE_DIST=1 GTOP_BENCH_REGION_REPEATS=5 $mode timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep "region repeat"
done; done > gpurun_out/r4_26_repeats.txt 2>&1
cat gpurun_out/r4_26_repeats.txt
