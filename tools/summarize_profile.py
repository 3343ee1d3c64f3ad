#!/usr/bin/env python3
"""Condense a rocprofv3 output directory (kernel_stats/kernel_trace/counter_collection CSVs under
gpurun_out/) into a small text summary suitable for profiles/ (kernel names shortened).
usage: tools/summarize_profile.py <rocprof_dir> <out.md> [title]"""
import collections
import csv
import glob
import os
import sys

d, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(d)


def short(n):
    n = n.replace("void pdwt::", "").replace("void ", "")
    return n.split("(")[0][:70]


lines = ["# %s" % title, ""]
for f in glob.glob(os.path.join(d, "*kernel_stats.csv")):
    lines += ["## rocprofv3 --kernel-trace --stats (per kernel, all launches)", "", "| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        lines.append("| %s | %s | %.1f | %.2f | %.2f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                                                 float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    lines.append("")
for f in glob.glob(os.path.join(d, "*kernel_trace.csv")):
    g = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f)):
        if "pdwt" not in r["Kernel_Name"]:
            continue
        k = (short(r["Kernel_Name"]), r["Grid_Size_X"] + "x" + r["Grid_Size_Y"])
        g[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size_X"])
    lines += ["## per (kernel, grid) from the kernel trace", "", "| kernel | grid (threads) | n | median us | min us | arch VGPR | acc VGPR | SGPR | LDS | wg |", "|---|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(g.items()):
        v = sorted(v)
        m = meta[k]
        lines.append("| %s | %s | %d | %.2f | %.2f | %s | %s | %s | %s | %s |" % (k[0], k[1], len(v), v[len(v) // 2] / 1e3, v[0] / 1e3, *m))
    lines.append("")
for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
    g = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "pdwt" not in r["Kernel_Name"]:
            continue
        g[(short(r["Kernel_Name"]), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines += ["## PMC counters (median per launch)", ""]
    for k, c in sorted(g.items()):
        lines.append("* `%s` grid %s: " % k + ", ".join("%s=%.4g" % (n, sorted(x)[len(x) // 2]) for n, x in sorted(c.items())))
    lines.append("")
open(out, "w").write("\n".join(lines) + "\n")
print("wrote", out)
