# round 4, call 32: the driver's N = 1 command three times after the rehearsal region went in for every run; multi tests
source tools/gpu_step.sh
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'], r['ms_per_step_gpu'], r['roofline']['frac'], r['config']['untimed_region_rehearsals'])"
done
step 900 gpurun_out/r4_32_tests.log python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu
tail -3 gpurun_out/r4_32_tests.log
