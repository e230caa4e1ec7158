#!/bin/bash
# PMC counters for the ESDF build kernels at 200^3 (one --pmc pass each)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_esdf
rm -rf $out; mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" ${ESDF_PMC_MORE:+"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"}; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 tools/esdf_time.py ${ESDF_GRID:-200} > $out/pass$i.log 2> $out/pass$i.err || echo "pass $i failed"
  echo "pass $i done: $ctrs"
done
python3 - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob("gpurun_out/pmc_esdf/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "esdf" in k:
            print(p.split("/")[2], k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
