# round 4, call 29: full GPU suite, smoke, the driver's N = 1 bench command
source tools/gpu_step.sh
step 1100 gpurun_out/r4_29_tests.log python3 -m pytest tests -x -q -m gpu
tail -4 gpurun_out/r4_29_tests.log
step 200 gpurun_out/r4_29_smoke.log python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
tail -2 gpurun_out/r4_29_smoke.log
step 300 gpurun_out/r4_29_bench.log python3 bench.py --gpus 1 --steps 20 --warmup 5
tail -1 gpurun_out/r4_29_bench.log | cut -c1-700
