source tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step 1100 gpurun_out/full_gpu_tests.log python -m pytest tests/ -q -m gpu || { tail -30 gpurun_out/full_gpu_tests.log; exit 1; }
tail -2 gpurun_out/full_gpu_tests.log
rm -rf gpurun_out/f4prof
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f4prof -- python3 tools/f4_time.py > gpurun_out/f4.log 2>&1
grep "edt_query\|trajectory_stats" gpurun_out/f4.log
step 600 gpurun_out/bench_driver.json python3 bench.py --gpus 1 --steps 20 --warmup 5
tail -1 gpurun_out/bench_driver.json | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['parity']['ok'])
for w in r['extras']['workloads']: print('  ', w['workload'][:50], round(w['us_per_launch'],1), round(w['roofline']['frac'],3), w['parity']['ok'])
print('  optimizer', r['extras']['optimizer'])
print('  cpu', r['cpu_baseline'])
"
