#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs written by tools/pmc_collect.sh: per-launch
averages for the evaluation kernel (gtop_eval_wave_kernel), plus the HBM-traffic figure used by bench.py's
roofline.traffic.

Traffic rule (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE tallies every fabric read request at 64 B although
128-B requests move 128 B, so the read side is rebuilt from the request-size
counters: 32*N32 + 64*N64 + 128*N128 (= FETCH_SIZE*1024 + 64*N128 when all
three are consistent).  WRITE_SIZE is taken as is.
usage: tools/pmc_summary.py gpurun_out/pmc_<tag> [min_grid_threads]"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    avg = {}
    import os
    passes = sorted(glob.glob(f"{d}/pass*/"))
    for pd in passes:
        cands = glob.glob(f"{pd}/**/*counter_collection.csv", recursive=True)
        if not cands:
            continue
        f = max(cands, key=os.path.getmtime)     # gpurun merges into an existing directory: take the newest run only
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gtop_eval" in r["Kernel_Name"] and int(r["Grid_Size"]) >= min_grid:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v) // 4:]          # drop the first quarter (warm-up / parity launches)
            avg[k] = sum(v) / len(v)
            avg["_n_" + k] = len(v)
    out = {"dir": d, "per_launch": {k: v for k, v in avg.items() if not k.startswith("_n_")}}
    if "FETCH_SIZE" in avg:
        fetch_raw = avg["FETCH_SIZE"] * 1024
        n32, n64, n128 = (avg.get(f"TCC_EA0_RDREQ_{s}_sum") for s in ("32B", "64B", "128B"))
        rd = None
        if None not in (n32, n64, n128):
            rd = 32 * n32 + 64 * n64 + 128 * n128
        wr = avg.get("WRITE_SIZE", 0.0) * 1024
        out["hbm"] = {"fetch_size_bytes_raw": fetch_raw, "read_bytes_from_request_sizes": rd,
                      "write_bytes": wr, "traffic_bytes": (rd if rd is not None else 2 * fetch_raw) + wr}
    if "TCC_HIT_sum" in avg:
        out["l2_hit_rate"] = avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
