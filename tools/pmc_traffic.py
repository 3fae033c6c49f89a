"""Aggregates gpurun_out/pmc_traffic/{FETCH_SIZE,WRITE_SIZE} (tools/pmc_traffic.sh) into profiles/<round>_pmc_traffic_cfg3.csv and
profiles/pmc_traffic.json (bench.py reads the latter for roofline.traffic); `bf16`: profiles/<round>_pmc_traffic_cfg5_bf16.{csv,json}.
Round tag: URSN_ROUND (default r03).  gfx950: FETCH_SIZE counts 64 B per 128-B
request for wide coalesced loads (MI355X_MICROARCH.md) -> hbm_bytes = 2 * fetch + write."""
import collections
import csv
import glob
import json
import re
import os
import sys

ROUND = os.environ.get("URSN_ROUND", "r04")
BF16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"   # python tools/pmc_traffic.py bf16 -> profiles/r02_pmc_traffic_cfg5_bf16.json

LABEL = [  # rocprof kernel name (regex) -> bench.py kernel label
    (r"twgradz_kernel<3", "twgradz<8,8>"),
    (r"tconv_kernel<8, 8, 3, false", "tconv<8,8>"), (r"tconv_kernel<8, 8, 3, true", "tconv_dgrad<8,8>"),
    (r"tconv_kernel<16, 16, 3, false", "tconv<16,16>"), (r"tconv_kernel<16, 16, 3, true", "tconv_dgrad<16,16>"),
    (r"twgrad_kernel<16, 16, 3>", "twgrad<16,16>"),
    (r"igemm_at_kernel<3, 32, 8, false", "igemm_at<32>"), (r"igemm_at_kernel<3, 32, 8, true", "igemm_at_dgrad<32>"),
    (r"igemm_at_kernel<3, 16, 16, false", "igemm_at<16>"), (r"igemm_wgrad_kernel<3, 2>", "igemm_wgrad<32>"),
    (r"s2conv_kernel<3", "s2conv"), (r"s2wgrad_kernel<3>", "s2wgrad"), (r"s2scatter_kernel<3", "s2scatter"),
    (r"dconv_kernel<27, true", "dconv"), (r"dconv_kernel<27, false, false", "dconv_dgrad"),
    (r"tconv_kernel<8, 8, 3, true, false, 1, true", "tconv_dgrad<8,8>+dz"),
    (r"tdeconv_kernel<16, 8, 3, true", "tdeconv<16,8>"), (r"tdeconv_kernel<16, 8, 3, false, true, true", "tdeconv<16,8>+pw"),
    (r"bn_bwd_apply_kernel", "bn_bwd_apply"), (r"bn_bwd_reduce_kernel", "bn_bwd_reduce"), (r"bn_act_kernel", "bn_act"),
]
if BF16:
    LABEL = [(r"b3conv_kernel<8, 8, false, false, 0, false", "b3conv_bf16<8,8>(dgrad)"), (r"b3conv_kernel<8, 8, true, false, 0, false", "b3conv_bf16<8,8>"),
             (r"b3conv_kernel<8, 8, true, false, 0, true", "b3conv_bf16<8,8>+bn"), (r"b3conv_kernel<16, 16, true, false, 0, false", "b3conv_bf16<16,16>"),
             (r"b3conv_kernel<16, 8, true", "b3conv_bf16<16,8>"), (r"b3conv_kernel<8, 16, false, true", "b3conv_bf16<8,16>+pw"),
             (r"b3wgradz_kernel<false>", "b3wgrad_bf16<8,8>(pair)"), (r"b3wgrad_kernel<16, 8", "b3wgrad_bf16<16,8>"),
             (r"bdeconv_kernel<true>", "bdeconv_bf16<16,8>"), (r"bpw_kernel<16, 8", "bpw_bf16"),
             (r"bcbconv_kernel<32, true>", "bcbconv_bf16<32>"), (r"bcbconv_kernel<32, false>", "bcbconv_bf16<32>(dgrad)"),
             (r"bcbconv_kernel<16, false>", "bcbconv_bf16<16>(dgrad)"), (r"b3conv_kernel<16, 16, false, false, 0, false", "b3conv_bf16<16,16>(dgrad)"),
             (r"bdconv_kernel<2, 8", "bdconv_bf16<2,8>"), (r"bdconv_kernel<4, 8", "bdconv_bf16<4,8>"), (r"bdconv_kernel<2, 4", "bdconv_bf16<2,4>+splitk"),
             (r"bbn_bwd_apply_kernel<true, 0, false, false", "bbn_bwd_apply(C8)"), (r"bbn_bwd_reduce_kernel<true, 0, false", "bbn_bwd_reduce(C8)"),
             (r"bbn_act_kernel<true, false, false, false", "bbn_act(C8)"),
             (r"bsconv_kernel<1, 8", "bsconv_bf16<1,8>"), (r"bsconv_kernel<2, 8", "bsconv_bf16<2,8>"), (r"bsconv_kernel<4, 4", "bsconv_bf16<4,4>"),
             (r"b0conv_kernel", "b0conv_bf16<1,8>"), (r"b0wgrad_kernel", "b0wgrad_bf16<1,8>"), (r"bbn_cat_kernel", "bbn_act(concat)"),
             (r"bhead_kernel", "bhead"), (r"bpack_multi_kernel", "bpack_multi")]

vals = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in vals:
    for f in glob.glob("gpurun_out/pmc_traffic/%s/*/*_counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                vals[c][r["Kernel_Name"]].append(float(r["Counter_Value"]))
rows, out = [], {"_note": ("cfg5_3d256_f8_b4_bf16" if BF16 else "cfg3_3d192_f8_b4") + ", rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/pmc_traffic.sh, "
                          "URSN_WGRAD_STREAM=0). Units KiB per dispatch averaged over all launches of the kernel. gfx950: FETCH_SIZE counts "
                          "64 B per 128-B request for wide coalesced loads (MI355X_MICROARCH.md, HBM) -> hbm_bytes_per_launch = 2*fetch + write."}
for k in sorted(vals["FETCH_SIZE"], key=lambda k: -sum(vals["FETCH_SIZE"][k])):
    f = vals["FETCH_SIZE"][k]
    w = vals["WRITE_SIZE"].get(k, [0.0])
    fa, wa = sum(f) / len(f), sum(w) / max(len(w), 1)
    rows.append((k, len(f), round(fa, 1), round(wa, 1)))
    for pat, lab in LABEL:
        if re.search(pat, k) and lab not in out:
            out[lab] = {"launches": len(f), "fetch_kib_raw": fa, "write_kib": wa, "hbm_bytes_per_launch": (2 * fa + wa) * 1024}
if BF16:   # bench.py's label "b3conv_bf16<8,8>" covers the plain forward and data-gradient instantiations: launch-weighted mean
    fs = [v for k in vals["FETCH_SIZE"] if re.search(r"b3conv_kernel<8, 8, (true|false), false, 0, false", k) for v in vals["FETCH_SIZE"][k]]
    ws = [v for k in vals["WRITE_SIZE"] if re.search(r"b3conv_kernel<8, 8, (true|false), false, 0, false", k) for v in vals["WRITE_SIZE"][k]]
    if fs and ws:
        fa, wa = sum(fs) / len(fs), sum(ws) / len(ws)
        out["b3conv_bf16<8,8>"] = {"launches": len(fs), "fetch_kib_raw": fa, "write_kib": wa, "hbm_bytes_per_launch": (2 * fa + wa) * 1024}
with open("profiles/%s_pmc_traffic_cfg5_bf16.csv" % ROUND if BF16 else "profiles/%s_pmc_traffic_cfg3.csv" % ROUND, "w") as fh:
    fh.write("kernel,launches,FETCH_SIZE_KiB_raw_per_launch,WRITE_SIZE_KiB_per_launch\n")
    for k, n, fa, wa in rows:
        fh.write('"%s",%d,%s,%s\n' % (k, n, fa, wa))
json.dump(out, open("profiles/%s_pmc_traffic_cfg5_bf16.json" % ROUND if BF16 else "profiles/pmc_traffic.json", "w"), indent=1)
for lab in out:
    if lab != "_note":
        print("%-22s %8.1f MB/launch" % (lab, out[lab]["hbm_bytes_per_launch"] / 1e6))
