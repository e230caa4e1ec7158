#!/usr/bin/env python3
"""Rebuild profiles/traffic.json (read by bench.py for roofline.traffic) from the
PMC summaries under profiles/r2/pmc/ (tools/pmc_collect.sh + tools/pmc_summary.py)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"B1024_m6_g200_f64": "summary_B1024_f64.json", "B16384_m6_g200_f64": "summary_B16384_f64.json",
        "B16384_m6_g200_f32": "summary_B16384_f32.json", "B8192_m12_g400_f64": "summary_B8192_m12_g400_f64.json"}
out = {}
for key, fn in KEYS.items():
    d = json.load(open(os.path.join(ROOT, "profiles", "r2", "pmc", fn)))
    h = d["hbm"]
    out[key] = {
        "traffic_bytes": h["traffic_bytes"], "read_bytes": h["read_bytes_from_request_sizes"],
        "write_bytes": h["write_bytes"], "fetch_size_raw_bytes": h["fetch_size_bytes_raw"],
        "l2_hit_rate": d.get("l2_hit_rate"),
        "source": f"profiles/r2/pmc/{fn} (rocprofv3 --pmc, one pass per counter group, tools/pmc_collect.sh; read side "
                  "rebuilt from TCC_EA0_RDREQ_{32B,64B,128B} because FETCH_SIZE tallies 128-B requests at 64 B on gfx950)",
    }
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["traffic_bytes"] / 1e6, 2) for k, v in out.items()}))
