#!/usr/bin/env python3
"""Print start / end (us, from the first listed kernel) of the launches [--skip, --skip + --n) of a rocprofv3 kernel trace whose name
contains one of the given substrings.

    python3 tools/trace_intervals.py <dir with *kernel_trace.csv> [--skip 40] [--n 12] k_stream k_frame_multi
"""
import argparse
import csv
import glob
import os

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--skip", type=int, default=40)
ap.add_argument("--n", type=int, default=12)
ap.add_argument("names", nargs="+")
a = ap.parse_args()
path = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))
             if any(n in r["Kernel_Name"] for n in a.names)), key=lambda t: t[0])
ev = ev[a.skip:a.skip + a.n]
t0 = ev[0][0]
for s, e, k in ev:
    print("%10.1f %10.1f %8.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k[:60]))
