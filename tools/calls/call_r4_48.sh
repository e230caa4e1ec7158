# round 4, call 48: the launch rule with one lane per segment past 12 segments: the whole GPU suite; the rule left to itself
source tools/gpu_step.sh
step 1100 gpurun_out/r4_48_tests.log python3 -m pytest tests -x -q -m gpu
tail -5 gpurun_out/r4_48_tests.log
timeout -k 10 400 python3 tools/variant_times_short.py 8192,13,f64 8192,24,f64 8192,32,f64 4096,48,f64 1024,13,f64 8192,13,f32 8192,7,f64 16384,3,f64 16384,6,f64 2>&1 | grep "B=" > gpurun_out/r4_48_times.txt
cat gpurun_out/r4_48_times.txt
