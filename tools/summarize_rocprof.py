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
    stats = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_ALL)
            w.writerow(rows[0])
            for r in rows[1:]:
                w.writerow([short(r[0])] + r[1:])
    acc = defaultdict(lambda: [0.0, 0])
    for path in sorted(glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
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
