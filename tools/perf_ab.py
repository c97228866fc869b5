#!/usr/bin/env python3
"""A/B timing of launch plans and kernel variants on one MI355X (run through gpurun).

    python tools/perf_ab.py [--steps 600] [--reps 3] case [case ...]

A case is  size:dtype:arith[:key=value,...]  e.g.  4096:f32:fast   4096:f32:strict:tb_steps=4   8192x1024:f64:fast:coll=SRT,turb=1
(keys: coll, turb, kernel, layout and every CavitySolver tuning switch).  Prints GLUPS (best of --reps timings of --steps steps
after a device wake-up and a warm-up) and microseconds per step, one line per case.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latticeboltzmannsimulations_amd import CavitySolver  # noqa: E402


def parse(case):
    parts = case.split(":")
    size, dt, arith = parts[0], parts[1], parts[2]
    nx, ny = (int(v) for v in size.split("x")) if "x" in size else (int(size), int(size))
    kw = dict(RT="MRT", turb=0, kernel="auto", layout="auto")
    tune = {}
    if len(parts) > 3 and parts[3]:
        for kv in parts[3].split(","):
            k, v = kv.split("=")
            if k == "coll":
                kw["RT"] = v
            elif k in ("kernel", "layout"):
                kw[k] = v
            elif k == "turb":
                kw["turb"] = int(v)
            elif k in ("tb_steps", "frame_seg"):
                tune[k] = int(v)
            else:
                tune[k] = v not in ("0", "false", "False")
    return nx, ny, np.float32 if dt == "f32" else np.float64, arith, kw, tune


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("cases", nargs="+")
    a = ap.parse_args()
    for case in a.cases:
        nx, ny, dtype, arith, kw, tune = parse(case)
        with CavitySolver(nx, ny, 1000.0, dtype=dtype, arith=arith, tuning=tune, **kw) as s:
            s.copy_bandwidth(1 << 30, 60)
            s.step(max(60, a.steps // 10)); s.sync()
            ms = min(s.time_steps(a.steps) for _ in range(a.reps)) / a.steps
            print(f"{case:48s} S={s.next_unit(1000)}  {nx * ny / ms / 1e6:8.1f} GLUPS  {ms * 1e3:9.2f} us/step", flush=True)


if __name__ == "__main__":
    main()
