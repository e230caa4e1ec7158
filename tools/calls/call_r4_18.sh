# round 4, call 18: window update eager against a replayed hipGraph of it
source tools/gpu_step.sh
step 300 gpurun_out/r4_18_window.log python3 tools/window_time.py 200 400
cat gpurun_out/r4_18_window.log
