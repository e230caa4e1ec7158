source tools/gpu_step.sh
step 600 gpurun_out/pytest18.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_analytic.py tests/test_golden.py -q -m gpu
tail -5 gpurun_out/pytest18.log
for lib in default NO6; do
  if [ $lib = default ]; then unset GTOP_HIP_LIB; else export GTOP_HIP_LIB=$PWD/build_var/lib$lib.so; fi
  for cfg in "--batch 16384 --dtype f32" "--batch 65536 --dtype f32" "--batch 8192 --dtype f32" "--batch 16384 --dtype f64 --spl 6 --waves 1" "--batch 8192 --segments 12 --grid 400 --density 0.04" "--batch 8192 --segments 12 --grid 200" "--batch 16384 --segments 10 --grid 200"; do
    step 200 gpurun_out/b_m.json python bench.py --no-extras --no-cpu-baseline --steps 300 $cfg
    python -c "
import json,sys
for l in open('gpurun_out/b_m.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$lib $cfg', '%.3f us' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], d['parity']['ok'], d['parity']['max_rel_grad'])
"
  done
done
