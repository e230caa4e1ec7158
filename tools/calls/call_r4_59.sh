# round 4, call 59: the optimizer loop, launch rule of before (pin 0) against 21 / m trajectories per wavefront (pin 10), by
# trajectory length and batch
source tools/gpu_step.sh
export GTOP_OPT_ROWS="0:f64,10:f64,0:f32,10:f32"
for M in 3 5 7 10; do
  GTOP_M=$M timeout -k 10 300 python3 tools/opt_time.py 4096 8192 16384 2>&1 | grep "B="
done > gpurun_out/r4_59_opt.txt 2>&1
cat gpurun_out/r4_59_opt.txt
