"""Group a rocprofv3 kernel trace by (kernel, grid) -> launches and time per step.
usage: trace_groups.py trace.csv steps [rows] [filter]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
flt = sys.argv[4] if len(sys.argv) > 4 else ""
g = defaultdict(list)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[-44:]
    if flt and flt not in name:
        continue
    g[(name, r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{sum(v) / steps:7.3f} ms/step  {len(v) / steps:5.1f}/step  avg {sum(v) / len(v):6.3f} max {max(v):6.3f}  grid {k[1]}x{k[2]}  {k[0]}")
print(f"total {sum(sum(v) for v in g.values()) / steps:.2f} ms/step")
