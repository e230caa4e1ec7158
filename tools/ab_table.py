#!/usr/bin/env python3
"""min us per (workload, library) from the output of tools/ab_bisect.sh"""
import collections
import re
import sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
lib = None
for line in open(sys.argv[1]):
    m = re.match(r"=== (\S+)", line)
    if m:
        lib = m.group(1)
        continue
    m = re.match(r"(B=.*spl=\d+):\s+([\d.]+) us", line)
    if m:
        rows[m.group(1)][lib].append(float(m.group(2)))
for k, v in rows.items():
    print(k, {a: min(b) for a, b in v.items()})
