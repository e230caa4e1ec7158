#!/bin/bash
: ${CFGS:="--batch,1024,--dtype,f64,--spl,1 --batch,1024,--dtype,f64,--spl,3 --batch,16384,--dtype,f64,--spl,3,--steps,500 --batch,16384,--dtype,f32,--spl,3,--steps,500 --batch,16384,--dtype,f32,--spl,1,--steps,500"}
# usage: tools/sweep.sh "<lib list>" -- runs the bench matrix for each kernel-variant library
for lib in $1; do
  if [ "$lib" = "default" ]; then unset GTOP_HIP_LIB; else export GTOP_HIP_LIB=$PWD/build_var/lib$lib.so; fi
  for cfg in $CFGS; do cfg=${cfg//,/ }
    timeout -k 5 100 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$cfg', '%.3e' % d['value'], '%.2f us' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], d['parity']['ok'])"
  done
done
