#!/usr/bin/env python3
"""Timeline of a few launch units from a rocprofv3 kernel trace: every kernel between the N-th and (N+count)-th launch of the bulk
kernel (k_stream / k_stepS_deep), times in us from the first.

    python3 tools/trace_units.py <dir with *kernel_trace.csv> [--unit 30] [--count 3]
"""
import argparse
import csv
import glob
import os

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--unit", type=int, default=30)
ap.add_argument("--count", type=int, default=3)
a = ap.parse_args()
path = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))), key=lambda t: t[0])
bulk = [i for i, e in enumerate(ev) if "k_stream" in e[2] or "k_stepS_deep" in e[2]]
i0, i1 = bulk[a.unit], bulk[a.unit + a.count]
t0 = ev[i0][0]
for s, e, k in ev[max(0, i0 - 3):i1 + 1]:
    print("%9.1f %9.1f %7.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k[:64]))
