"""Attributes kernel time to the roctx ranges of a `URSN_ROCTX=1 rocprofv3 --kernel-trace --hip-trace --marker-trace` run
(tools/profile_round.sh): every launch group of the plan pushes "UResNet/<scope>:<pass>" (csrc/conv_api.hip ursn_roctx_push), the
stand-in for the per-layer report of lib/ssnet_trainval.py:207-227.  Ranges are HOST intervals and the device runs far behind the
host, so kernels are matched through the launch call: a kernel-trace row carries the Correlation_Id of its hipLaunchKernel call, the
HIP API trace gives that call's host time, and the call belongs to the range open at that time.  Output: one CSV row per
(scope, pass) with launches, kernel names and summed device time.
    python tools/marker_attrib.py <dir> > profiles/<round>_roctx_attribution_cfg3.csv"""
import bisect
import collections
import csv
import glob
import sys

d = sys.argv[1]


def one(pat):
    f = sorted(glob.glob(d + "/*/" + pat) + glob.glob(d + "/" + pat))
    if not f:
        sys.exit("no %s under %s" % (pat, d))
    return f[0]


ranges = []
for r in csv.DictReader(open(one("*marker_api_trace.csv"))):
    if r["Function"].startswith("UResNet/"):
        ranges.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
ranges.sort()
starts = [r[0] for r in ranges]
launch_host = {}   # correlation id of a launch call -> host start time
for r in csv.DictReader(open(one("*hip_api_trace.csv"))):
    if "Launch" in r["Function"]:
        launch_host[r["Correlation_Id"]] = int(r["Start_Timestamp"])
agg, outside = collections.OrderedDict(), [0, 0.0]
for r in csv.DictReader(open(one("*kernel_trace.csv"))):
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    t = launch_host.get(r["Correlation_Id"])
    i = bisect.bisect_right(starts, t) - 1 if t is not None else -1
    if i < 0 or t > ranges[i][1]:
        outside[0] += 1; outside[1] += ms   # head, Adam, zero_gradients, torch kernels: launched outside every range
        continue
    e = agg.setdefault(ranges[i][2], [0, 0.0, collections.Counter()])
    e[0] += 1
    e[1] += ms
    e[2][r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:48]] += 1
w = csv.writer(sys.stdout)
w.writerow(["scope:pass", "kernel_launches", "device_ms_total", "kernels"])
for k, (n, ms, names) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k, n, "%.3f" % ms, "; ".join("%s x%d" % kv for kv in names.most_common(4))])
w.writerow(["(outside every range)", outside[0], "%.3f" % outside[1], ""])
