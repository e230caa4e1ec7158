# round 3: fabric-side counters of edt_query_kernel (2^20 random queries, static only and with 32 boxes: tools/f4_time.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_edt; rm -rf $out; mkdir -p $out
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 tools/f4_time.py > $out/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_edt/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "edt_query_kernel<false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
# launches alternate: 11 with boxes first, then 11 static only (tools/f4_time.py)
for name, sl in (("32 boxes", slice(0, 11)), ("static only", slice(11, 22))):
    print(name, {k: round(sum(v[sl]) / max(1, len(v[sl])), 1) for k, v in sorted(acc.items())})
PY
