source tools/gpu_step.sh
for B in 1 16 64 256; do for spl in 0 3; do
  step 120 gpurun_out/b_s.json python bench.py --no-extras --no-cpu-baseline --steps 1000 --batch $B --spl $spl
  python -c "
import json
for l in open('gpurun_out/b_s.json'):
    if l.startswith('{'):
        d=json.loads(l); print('B=$B spl=$spl', '%.3f us' % d['roofline']['avg_launch_us'], d['parity']['ok'])
"
done; done
SPL=0 python tools/host_api_rate.py 1 2>/dev/null | head -2
SPL=3 python tools/host_api_rate.py 1 2>/dev/null | head -2
