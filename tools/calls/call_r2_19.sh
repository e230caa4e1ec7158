source tools/gpu_step.sh
for i in 1 2; do
step 300 gpurun_out/bench_driver.json python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline
python - <<'PY'
import json
for l in open('gpurun_out/bench_driver.json'):
    if l.startswith('{'):
        d=json.loads(l); print('driver-style value %.4g ms/step %.5f launch %.3f (timed region %.3f) frac %.3f buckets of %d' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['avg_launch_us_timed_region'], d['roofline']['frac'], d['config']['steps_per_bucket']))
PY
done
