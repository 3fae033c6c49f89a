"""Aggregates rocprofv3 --pmc csv output (gpurun_out/pmc_<TAG>/*/...) per kernel: python tools/pmc_report.py TAG [filter]"""
import collections
import csv
import glob
import sys

tag = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_%s/*/*/*_counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if flt and flt not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    c = {n: v / cnt[(k, n)] for n, v in agg[k].items()}
    print(k)
    wc = c.get("SQ_WAVE_CYCLES", 0)
    simd_cycles = c.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024
    for n in sorted(c):
        extra = ""
        if n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") and wc:
            extra = "  %.0f%% of wave cycles" % (100 * c[n] / wc)
        if n == "SQ_VALU_MFMA_BUSY_CYCLES" and simd_cycles:
            extra = "  %.0f%% of SIMD cycles" % (100 * c[n] / simd_cycles)
        if n == "SQ_LDS_BANK_CONFLICT" and c.get("SQ_LDS_IDX_ACTIVE"):
            extra = "  %.0f%% of LDS cycles" % (100 * c[n] / c["SQ_LDS_IDX_ACTIVE"])
        print("   %-28s %16.0f%s" % (n, c[n], extra))
