"""Group a rocprofv3 kernel trace by (kernel, grid) -> launches and time per step.  usage: trace_groups.py trace.csv steps"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
g = defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r.get("LDS_Block_Size", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    g[key][0] += 1
    g[key][1] += d
tot = sum(v[1] for v in g.values())
print(f"total {tot / steps:.2f} ms/step")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{v[1] / steps:8.3f} ms/step {v[0] / steps:6.1f} launches/step  {v[1] / v[0]:8.3f} ms each  grid {k[1]}x{k[2]} lds {k[3]}  {k[0]}")
