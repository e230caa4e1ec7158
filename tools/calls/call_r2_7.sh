source tools/gpu_step.sh
step 900 gpurun_out/pytest7.log python -m pytest tests -q -m gpu
tail -4 gpurun_out/pytest7.log
step 300 gpurun_out/bench_driver.json python bench.py --gpus 1 --steps 20 --warmup 5
python - <<'PY'
import json
for l in open('gpurun_out/bench_driver.json'):
    if l.startswith('{'):
        d=json.loads(l); print('driver-style', d['value'], d['ms_per_step'], d['warmup'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
        for w in d['extras']['workloads']: print(w['workload'], w.get('us_per_launch'), w.get('roofline',{}).get('frac'), w['parity']['ok'])
        print(d['extras'].get('optimizer'))
PY
step 120 gpurun_out/b_1024.json python bench.py --no-extras --no-cpu-baseline
grep -o '"avg_launch_us": [0-9.]*\|"frac": [0-9.]*\|"value": [0-9.]*' gpurun_out/b_1024.json
