source tools/gpu_step.sh
step 600 gpurun_out/pytest17.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_analytic.py -q -m gpu -x
tail -3 gpurun_out/pytest17.log
for dt in f64 f32; do
for B in 3072 4096 8192 16384 65536; do
for spl in 3 6; do
  step 120 gpurun_out/b_m.json python bench.py --no-extras --no-cpu-baseline --steps 300 --batch $B --dtype $dt --spl $spl --waves 1
  python -c "
import json,sys
for l in open('gpurun_out/b_m.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$dt B=$B spl=$spl', '%.3f us' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], d['parity']['ok'])
"
done; done; done
