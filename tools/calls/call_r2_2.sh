source tools/gpu_step.sh
step 600 gpurun_out/pytest2.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_multi.py -q -m gpu -x
tail -5 gpurun_out/pytest2.log
for lib in default W0; do
  for B in 1024 2048 4096; do
    if [ $lib = default ]; then unset GTOP_HIP_LIB; else export GTOP_HIP_LIB=$PWD/build_var/lib$lib.so; fi
    step 120 gpurun_out/b_${lib}_$B.json python bench.py --no-extras --no-cpu-baseline --steps 1000 --batch $B
    python -c "
import json,sys
for l in open('gpurun_out/b_${lib}_$B.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$lib', $B, '%.3f us' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], 'value %.3e' % d['value'], d['parity'])
"
  done
done
