"""Largest launches per kernel name in a rocprofv3 kernel trace.  usage: trace_top.py trace.csv [filter]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
g = defaultdict(list)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-50:]
    if flt and flt not in name:
        continue
    g[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for name, d in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    d.sort(reverse=True)
    print(f"{name:45s} n={len(d):5d} total {sum(d):9.2f} ms  top: " + " ".join(f"{x:.3f}" for x in d[:24:2]))
