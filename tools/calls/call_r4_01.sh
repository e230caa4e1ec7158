# round 4, call 1: the boundary-hardening batch and the new bench line (device-clock stamps, clock warm-up before the region,
# one graph at N = 1) on the round-3 kernels — the baseline every later kernel change is compared with
source tools/gpu_step.sh
step 600 gpurun_out/r4_01_tests.log python3 -m pytest tests/test_capi.py tests/test_rendezvous.py tests/test_cpp_shim.py tests/test_gpu_multi.py -x -q -m gpu
tail -3 gpurun_out/r4_01_tests.log
step 300 gpurun_out/r4_01_bench_driver.log python3 bench.py --gpus 1 --steps 20 --warmup 5
grep '^{' gpurun_out/r4_01_bench_driver.log > gpurun_out/r4_01_bench_driver.json
step 200 gpurun_out/r4_01_bench_2000.log python3 bench.py --no-extras --no-cpu-baseline
grep '^{' gpurun_out/r4_01_bench_2000.log > gpurun_out/r4_01_bench_2000.json
python3 - <<'PY'
import json
for f in ("r4_01_bench_driver", "r4_01_bench_2000"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    print(f, "value %.4g ms/step host %.5f gpu %.5f" % (d["value"], d["ms_per_step"], d["ms_per_step_gpu"]), d["roofline"]["launch_us"], d["roofline"]["frac_by_source"], d["config"]["clock_warmup_steps"])
PY
