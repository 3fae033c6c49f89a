"""Per-kernel LDS activity from tools/pmc_lds.sh: LDS-busy share of the kernel's cycles and the share of that lost to bank
conflicts.  SQ_LDS_* are summed over the CUs; GRBM_GUI_ACTIVE over the 8 XCDs."""
import collections, csv, glob, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_lds/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, c in agg.items():
    if c.get("SQ_LDS_IDX_ACTIVE", 0) <= 0 or c.get("GRBM_GUI_ACTIVE", 0) <= 0:
        continue
    cu_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 256.0
    out[k] = {"lds_active_frac_of_cu_cycles": round(c["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 3),
              "bank_conflict_frac_of_lds_active": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3),
              "bank_conflict_frac_of_cu_cycles": round(c["SQ_LDS_BANK_CONFLICT"] / cu_cycles, 3),
              "gpu_cycles_total": round(c["GRBM_GUI_ACTIVE"] / 8.0)}
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["gpu_cycles_total"])), sys.stdout, indent=1)
