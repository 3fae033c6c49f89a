"""Per-kernel MFMA-busy fraction from the rocprofv3 --pmc pass of tools/pmc_mfma.sh (north_star: 'rocprof ... MFMA-busy').
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs; the available SIMD cycles of a dispatch are
GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 x 1024 SIMDs (MI355X_MICROARCH.md, cycle constants / DVFS note).
SQ_WAVE_CYCLES, SQ_ACTIVE_INST_ANY, SQ_WAIT_* count quad-cycles per wave."""
import collections, csv, glob, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_mfma/*counter_collection.csv") + glob.glob("gpurun_out/pmc_mfma/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k)); n[k] += 1
out = {}
for k, c in agg.items():
    simd = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if not simd or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) <= 0:
        continue
    out[k] = {"dispatches": n[k], "mfma_busy_frac_of_simd_cycles": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4),
              "active_inst_frac_of_wave_cycles": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4) if wc else None,
              "wait_any_frac_of_wave_cycles": round(c.get("SQ_WAIT_ANY", 0) / wc, 4) if wc else None,
              "wait_inst_frac_of_wave_cycles": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 4) if wc else None,
              "gpu_active_cycles_per_dispatch": round(c.get("GRBM_GUI_ACTIVE", 0) / 8.0 / n[k])}
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_frac_of_simd_cycles"])), sys.stdout, indent=1)
