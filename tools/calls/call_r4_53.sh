# round 4, call 53: rocprofv3 kernel stats of a three-lanes-per-segment workload (8 192 trajectories of 10 segments) and of a
# one-lane-per-segment one (8 192 of 13 segments); then the whole GPU suite once more
source tools/gpu_step.sh
for w in "8192 10" "8192 13"; do set -- $w
  rm -rf gpurun_out/kprof_m$2
  step 300 gpurun_out/kprof_m$2.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_m$2 -- python3 bench.py --no-extras --no-cpu-baseline --batch $1 --segments $2 --steps 300
  find gpurun_out/kprof_m$2 -name '*kernel_trace.csv' -delete
  f=$(ls -t gpurun_out/kprof_m$2/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_B$1_m$2_f64.csv; head -2 $f | tail -1 | cut -c1-120,330-
  grep '^{' gpurun_out/kprof_m$2.log | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['config']['workload'], r['ms_per_step_gpu'], r['roofline']['frac'], r['roofline']['kernel'][:70], r['parity'])"
done
step 1100 gpurun_out/r4_53_tests.log python3 -m pytest tests -x -q -m gpu
tail -3 gpurun_out/r4_53_tests.log
