# round 4, call 11: the GPU suite after the group window entry + precision knob + 16-byte fp32 record stores, smoke,
# the driver's bench command, and the optimizer's wall time at large batches (is two trajectories per wavefront worth it?)
source tools/gpu_step.sh
step 1100 gpurun_out/r4_11_tests.log python3 -m pytest tests -x -q -m gpu
tail -4 gpurun_out/r4_11_tests.log
step 200 gpurun_out/r4_11_smoke.log python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
tail -2 gpurun_out/r4_11_smoke.log
step 300 gpurun_out/r4_11_bench.log python3 bench.py --gpus 1 --steps 20 --warmup 5
tail -1 gpurun_out/r4_11_bench.log | cut -c1-1500
step 300 gpurun_out/r4_11_opt_time.log python3 tools/opt_time.py 1024 4096 16384
cat gpurun_out/r4_11_opt_time.log
