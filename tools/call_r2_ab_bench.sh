#!/bin/bash
# A/B of two libraries on one box: alternating bench runs (B=1024 f64 default), value + kernel us per run
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for v in $A $B; do
    lib=$GRAFT_REPO_ROOT/grad_traj_optimization_amd/libgtop_$v.so
    GTOP_HIP_LIB=$lib timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v failed"; tail -5 gpurun_out/ab_$v.err; exit 1; }
    python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
r = json.loads([l for l in open(f"gpurun_out/ab_{v}.json") if l.startswith("{")][-1])
print(f"{v:6s} value {r['value']:.4g} ms/step {r['ms_per_step']*1e3:.3f} us  roofline launch {r['roofline'].get('avg_launch_us', r['roofline'].get('launch_us', 0))} frac {r['roofline']['frac']:.3f} parity {r['parity']['ok']}")
PY
  done
done
