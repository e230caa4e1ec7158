#!/usr/bin/env python3
"""Rebuild profiles/traffic.json — what bench.py reads for roofline.traffic, roofline.launch_us.rocprof and
roofline_issue — from the round's committed rocprofv3 outputs under profiles/r4/: the PMC summaries
(tools/pmc_collect.sh + tools/pmc_summary.py: fabric traffic, issued instructions per wavefront) and the
--kernel-trace --stats CSVs of the same bench commands (average launch time of the evaluation kernel)."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = "r4"
KEYS = {"B1024_m6_g200_f64": ("summary_B1024_f64.json", "kernel_stats_default_B1024_f64.csv"),
        "B16384_m6_g200_f64": ("summary_B16384_f64.json", "kernel_stats_B16384_f64.csv"),
        "B16384_m6_g200_f32": ("summary_B16384_f32.json", "kernel_stats_B16384_f32.csv"),
        "B8192_m12_g400_f64": ("summary_B8192_m12_g400_f64.json", "kernel_stats_B8192_m12_g400_f64.csv")}
out = {}
for key, (fn, ks) in KEYS.items():
    path = os.path.join(ROOT, "profiles", RND, "pmc", fn)
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    h, pl = d["hbm"], d["per_launch"]
    waves = pl["SQ_WAVES"]
    ipw = {k: pl.get("SQ_INSTS_" + n, 0.0) / waves for k, n in
           (("valu", "VALU"), ("salu", "SALU"), ("lds", "LDS"), ("vmem_rd", "VMEM_RD"), ("vmem_wr", "VMEM_WR"), ("smem", "SMEM"))}
    ipw["total"] = sum(ipw.values())
    entry = {
        "traffic_bytes": h["traffic_bytes"], "read_bytes": h["read_bytes_from_request_sizes"],
        "write_bytes": h["write_bytes"], "fetch_size_raw_bytes": h["fetch_size_bytes_raw"],
        "l2_hit_rate": d.get("l2_hit_rate"),
        "source": f"profiles/{RND}/pmc/{fn} (rocprofv3 --pmc, one pass per counter group, tools/pmc_collect.sh; read side "
                  "rebuilt from TCC_EA0_RDREQ_{32B,64B,128B} because FETCH_SIZE tallies 128-B requests at 64 B on gfx950)",
        "issued_per_wave": ipw, "waves_per_launch": waves,
        "lds_bank_conflict_frac": pl["SQ_LDS_BANK_CONFLICT"] / pl["SQ_LDS_IDX_ACTIVE"] if pl.get("SQ_LDS_IDX_ACTIVE") else None,
        "issue_source": f"profiles/{RND}/pmc/{fn}: SQ_INSTS_{{VALU,SALU,LDS,VMEM_RD,VMEM_WR,SMEM}} / SQ_WAVES",
    }
    kpath = os.path.join(ROOT, "profiles", RND, ks)
    if os.path.exists(kpath):
        for r in csv.DictReader(open(kpath)):
            if "gtop_eval" in r["Name"]:
                entry["rocprof_avg_us"] = float(r["AverageNs"]) * 1e-3
                entry["rocprof_kernel"] = r["Name"].split("(double const*")[0].split("(float const*")[0].replace("void ", "")
                entry["rocprof_source"] = (f"profiles/{RND}/{ks} (rocprofv3 --kernel-trace --stats of the same bench command; "
                                           f"{r['Calls']} launches)")
                break
    out[key] = entry
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: (round(v["traffic_bytes"] / 1e6, 2), round(v["issued_per_wave"]["total"]), v.get("rocprof_avg_us"))
                  for k, v in out.items()}))
