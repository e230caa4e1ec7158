#!/bin/bash
# instruction counters of one workload under several builds of the library (A/B): tools/pmc_ab.sh "<rows>" lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROWS=$1; shift
for L in "$@"; do
  tag=$(basename $L .so)
  out=gpurun_out/pmcab_$tag
  rm -rf $out; mkdir -p $out
  export GTOP_HIP_LIB=$(realpath $L)
  timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out -- python3 tools/variant_times_short.py $ROWS > $out/log.txt 2>&1
  python3 - "$out" "$tag" <<'PY'
import csv,glob,sys,collections
d,tag=sys.argv[1],sys.argv[2]
f=max(glob.glob(d+"/**/*counter_collection.csv",recursive=True))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "gtop_eval" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:90],r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    w=sum(v["SQ_WAVES"])/len(v["SQ_WAVES"])
    print(tag,k[1],{c:round(sum(x)/len(x)/w,1) for c,x in v.items() if c!="SQ_WAVES"},"waves",w,k[0][22:80])
PY
done
