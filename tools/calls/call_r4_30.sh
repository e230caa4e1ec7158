# round 4, call 30: rocprofv3 kernel stats of the batched optimizer (tools/opt_time.py: one launch = a whole optimisation)
# and of the gather kernel in the rank path at RCCL world size 1
source tools/gpu_step.sh
rm -rf gpurun_out/optprof gpurun_out/pushprof
step 300 gpurun_out/r4_30_opt.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/optprof -- python3 tools/opt_time.py 1024 16384
f=$(ls -t gpurun_out/optprof/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_optimizer.csv
find gpurun_out/optprof -name '*kernel_trace.csv' -delete
grep "B=" gpurun_out/r4_30_opt.log > gpurun_out/r4_30_opt_times.txt
head -12 gpurun_out/kernel_stats_optimizer.csv | cut -c1-260
export GTOP_BENCH_FORCE_DIST=1
step 300 gpurun_out/r4_30_push.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pushprof -- python3 bench.py --gpus 1 --steps 200 --warmup 5 --no-extras --no-cpu-baseline --gather-grads
f=$(ls -t gpurun_out/pushprof/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_rank_path_push.csv
find gpurun_out/pushprof -name '*kernel_trace.csv' -delete
head -6 gpurun_out/kernel_stats_rank_path_push.csv | cut -c1-260
