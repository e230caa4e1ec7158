source tools/gpu_step.sh
step 900 gpurun_out/pytest1.log python -m pytest tests -q -m gpu
tail -5 gpurun_out/pytest1.log
step 300 gpurun_out/bench_driver.json python bench.py --gpus 1 --steps 20 --warmup 5
step 200 gpurun_out/stamps_1024.txt env GTOP_HIP_LIB=$PWD/build_var/libS.so python tools/stamps.py 1024 3 1
cat gpurun_out/stamps_1024.txt
step 120 gpurun_out/prof_floor.log rocprofv3 --kernel-trace --stats -d gpurun_out/prof_floor -- tools/ubench/launch_floor2
step 300 gpurun_out/prof_b1024.log rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1024 -- python3 bench.py --no-extras --no-cpu-baseline --steps 500
find gpurun_out/prof_floor gpurun_out/prof_b1024 -name "*kernel_stats.csv" | xargs head -5
