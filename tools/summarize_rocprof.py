#!/usr/bin/env python3
"""Condense the rocprofv3 output of scratch/prof.sh-style runs (one --kernel-trace --stats pass and separate --pmc passes,
as the MI355X guide prescribes) into the two small files kept under profiles/:

    <prefix>_kernel_stats.csv   rocprofv3's own per-kernel statistics (copied, template arguments shortened)
    <prefix>_pmc_summary.csv    mean counter value per launch and kernel

usage: summarize_rocprof.py <rocprof dir> <profiles/prefix> ["header comment"]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main(src, prefix, comment=""):
    # Per-kernel statistics, recomputed from the kernel trace in rocprofv3's own column layout: the one-workgroup warm-up launch of
    # the streaming kernel that lbm_create issues (it returns at once: 0.6 us) is left out -- in rocprofv3's --stats table it would
    # count as a launch and pull the average down.
    trace = glob.glob(os.path.join(src, "trace", "*kernel_trace.csv"))
    if trace:
        dur = defaultdict(list)
        for r in csv.DictReader(open(trace[0])):
            if "k_stream" in r["Kernel_Name"] and r.get("Grid_Size_X") == r.get("Workgroup_Size_X"):
                continue
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        total = sum(sum(v) for v in dur.values()) or 1
        with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_ALL)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                mean = sum(v) / len(v)
                sd = (sum((x - mean) ** 2 for x in v) / (len(v) - 1)) ** 0.5 if len(v) > 1 else 0.0
                w.writerow([k, len(v), sum(v), "%.6f" % mean, "%.2f" % (100.0 * sum(v) / total), min(v), max(v), "%.6f" % sd])
    acc = defaultdict(lambda: [0.0, 0])
    for path in sorted(glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            if "k_stream" in r["Kernel_Name"] and str(r.get("Grid_Size")) == str(r.get("Workgroup_Size")):
                continue                      # (the warm-up launch, as above)
            key = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[key][0] += float(r["Counter_Value"])
            acc[key][1] += 1
    with open(prefix + "_pmc_summary.csv", "w", newline="") as f:
        if comment:
            f.write("# " + comment + "\n")
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
        for (k, c), (tot, n) in sorted(acc.items()):
            w.writerow([k, c, round(tot / n, 1), n])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
