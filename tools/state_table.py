"""Current-state table for DESIGN.md from `bench.py --layers --breakdown` stderr logs: per kernel label its launches per step,
ms per step, the fraction of its own roofline (sum of max(F / P, B / BW) over its launches / its measured time) and the achieved
rates.  usage: python tools/state_table.py profiles/r04_layers_cfg3.log [profiles/r04_pmc_mfma_busy_cfg3.json]"""
import collections
import json
import re
import sys

rows = collections.OrderedDict()
pat = re.compile(r"^(UResNet/\S+)\s+(fwd|dgrad|wgrad|bn_stats|bn_act|bn_bwd|head)\s+(.+?)\s+([\d.]+) ms\s+([\d.]+) TFLOP/s\s+([\d.]+) GB/s\s+(\d+)%")
for line in open(sys.argv[1]):
    m = pat.match(line)
    if not m:
        continue
    lname, ps, k, ms, tf, gb, pct = m.groups()
    ms, tf, gb, pct = float(ms), float(tf), float(gb), float(pct)
    r = rows.setdefault(k.strip(), {"n": 0, "ms": 0.0, "roof": 0.0, "flop": 0.0, "byte": 0.0})
    r["n"] += 1
    r["ms"] += ms
    r["roof"] += ms * pct / 100.0
    r["flop"] += tf * ms
    r["byte"] += gb * ms
busy = {}
if len(sys.argv) > 2:
    for k, v in json.load(open(sys.argv[2])).items():
        busy[k] = v["mfma_busy_frac_of_simd_cycles"]
tot = sum(r["ms"] for r in rows.values())
print("| kernel (bench label) | launches / step | ms / step | share | fraction of own roofline | achieved |")
print("|---|---|---|---|---|---|")
for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
    if r["ms"] < 0.05:
        continue
    ach = "%.0f TFLOP/s" % (r["flop"] / r["ms"]) if r["flop"] > 0 else "%.2f TB/s" % (r["byte"] / r["ms"] / 1e3)
    print("| `%s` | %d | %.2f | %.1f %% | %.2f | %s |" % (k, r["n"], r["ms"], 100 * r["ms"] / tot, r["roof"] / r["ms"], ach))
print("\nsum of timed launches: %.1f ms / step (weight-gradient stream serialised)" % tot)
