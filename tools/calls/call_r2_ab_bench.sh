#!/bin/bash
# A/B/... of libraries on one box: alternating bench runs, value + kernel us per run
# usage: call_r2_ab_bench.sh "hip varA varB" <bench args...>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
VARS=$1; shift
for i in 1 2 3; do
  for v in $VARS; do
    lib=$GRAFT_REPO_ROOT/grad_traj_optimization_amd/libgtop_$v.so
    GTOP_HIP_LIB=$lib timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v failed"; tail -5 gpurun_out/ab_$v.err; exit 1; }
    python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
r = json.loads([l for l in open(f"gpurun_out/ab_{v}.json") if l.startswith("{")][-1])
print(f"{v:6s} value {r['value']:.4g} us/step {r['ms_per_step']*1e3:.3f}  kernel {r['roofline'].get('avg_launch_us', 0):.3f} us frac {r['roofline']['frac']:.3f} parity {r['parity']['ok']}")
PY
  done
done
