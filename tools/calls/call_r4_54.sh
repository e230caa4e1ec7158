# round 4, call 54: final check — smoke and the driver's N = 1 command on the final build
source tools/gpu_step.sh
step 200 gpurun_out/r4_54_smoke.log python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
tail -2 gpurun_out/r4_54_smoke.log
step 400 gpurun_out/r4_54_bench.log python3 bench.py --gpus 1 --steps 20 --warmup 5
tail -1 gpurun_out/r4_54_bench.log | cut -c1-420
