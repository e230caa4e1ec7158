source tools/gpu_step.sh
step 200 gpurun_out/stamps_wave_1024.txt env GTOP_HIP_LIB=$PWD/build_var/libS.so python tools/stamps_wave.py 1024
cat gpurun_out/stamps_wave_1024.txt
step 100 gpurun_out/b_S.json env GTOP_HIP_LIB=$PWD/build_var/libS.so python bench.py --no-extras --no-cpu-baseline --steps 1000
grep -o '"avg_launch_us": [0-9.]*' gpurun_out/b_S.json
rocm-smi --showclocks 2>/dev/null | head -20
