# round 4, call 33: the push kernel takes the bucket's closing stamp itself; the region loop replays the graphs directly
source tools/gpu_step.sh
step 900 gpurun_out/r4_33_tests.log python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_api.py -x -q -m gpu
tail -3 gpurun_out/r4_33_tests.log
export GTOP_BENCH_REGION_SPLIT=1 GTOP_BENCH_REGION_REPEATS=3 GTOP_BENCH_FORCE_DIST=1
for mode in push library; do
  echo "=== $mode"
  GTOP_BENCH_GATHER=$mode timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep "region split"
done > gpurun_out/r4_33_split.txt 2>&1
cat gpurun_out/r4_33_split.txt
