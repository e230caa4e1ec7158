#!/usr/bin/env python3
"""Count instructions per region of gtop_eval_wave_kernel in a -DGTOP_MARKS assembly listing.
usage: make -C grad_traj_optimization_amd/csrc asm EXTRA=-DGTOP_MARKS && python tools/isa_regions.py [/tmp/gtop_kernels.s] [kernel-substring]"""
import re
import sys
from collections import Counter

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/gtop_kernels.s"
want = sys.argv[2] if len(sys.argv) > 2 else "gtop_eval_wave_kernelIdLb0ELi3ELi1E"
NAMES = {0: "kernel-argument fetch", 1: "group/lane indices, input addresses + loads", 2: "constants, free-variable offsets",
         3: "coefficients (A^-1 d)", 4: "sample times (+ tiny-T replay code)", 5: "stage A: pos/vel, indices, corner loads",
         6: "in flight: jerk term, speeds", 7: "stage B: blend, penalty, accumulation", 8: "A^-T",
         9: "tile writes", 10: "cost (DPP sum, store)", 11: "tile reads, sums, gradient store"}
inside, region = False, 0
cnt = {}
for line in open(path):
    if not inside:
        if line.startswith("_ZN") and want in line and line.rstrip().endswith(":") or (want in line and line.startswith("_ZN") and ": " in line):
            inside, region = True, 0
        continue
    m = re.search(r"GTOP_MARK (\d+)", line)
    if m:
        region = int(m.group(1))
        continue
    t = line.strip()
    if not t or t.startswith((";", ".", "_")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else \
        "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"
    cnt.setdefault(region, Counter())[kind] += 1
    if op == "s_endpgm":
        break
tot = Counter()
print(f"{'region':52s} {'valu':>5s} {'salu':>5s} {'lds':>4s} {'vmem':>5s} {'all':>5s}")
for r in sorted(cnt):
    c = cnt[r]
    tot += c
    print(f"{r:2d} {NAMES.get(r, ''):49s} {c['valu']:5d} {c['salu']:5d} {c['lds']:4d} {c['vmem']:5d} {sum(c.values()):5d}")
print(f"{'total':52s} {tot['valu']:5d} {tot['salu']:5d} {tot['lds']:4d} {tot['vmem']:5d} {sum(tot.values()):5d}")
