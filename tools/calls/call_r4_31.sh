# round 4, call 31: where the rank path's extra ~13 us per 20-step region (against the single path) go
source tools/gpu_step.sh
export GTOP_BENCH_REGION_SPLIT=1 GTOP_BENCH_REGION_REPEATS=3
for mode in single dist; do
  echo "=== $mode"
  if [ $mode = dist ]; then export GTOP_BENCH_FORCE_DIST=1; fi
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep "region split"
done > gpurun_out/r4_31_split.txt 2>&1
cat gpurun_out/r4_31_split.txt
