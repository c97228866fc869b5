#!/usr/bin/env python3
"""Enter the HBM bytes per launch of the dominant kernel, from a PMC summary written by tools/profile_bench.sh, into
profiles/traffic.json (what bench.py reports as roofline.traffic when the run's configuration, kernel and steps per launch match).

    python3 tools/update_traffic.py <prefix> <config:world:kernel:arith> <kernel substring> <steps per launch> <algorithmic bytes per step> "<bench args>"

bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024 per launch: the counters are in KiB, and on gfx950 FETCH_SIZE counts 64-byte
requests as 32 (the correction /opt/skills/guides/MI355X_MICROARCH.md prescribes).
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix, key, kern, S, alg, args = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
path = os.path.join(ROOT, "profiles", prefix + "_pmc_summary.csv")
vals = {}
for r in csv.reader(l for l in open(path) if not l.startswith("#")):
    if len(r) >= 3 and kern in r[0] and r[1] in ("FETCH_SIZE", "WRITE_SIZE"):
        vals[r[1]] = float(r[2])
fetch, write = int(vals["FETCH_SIZE"] * 1024 * 2), int(vals["WRITE_SIZE"] * 1024)
tpath = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tpath))
t[key] = {"hbm_bytes_per_launch": fetch + write, "steps_per_launch": S, "kernel": kern.split("<")[0], "fetch_bytes": fetch, "write_bytes": write,
          "algorithmic_bytes_per_step": alg,
          "source": f"profiles/{prefix}_pmc_summary.csv (tools/profile_bench.sh {prefix} {args}): {kern} per launch; bytes = FETCH_SIZE*1024*2 + "
                    "WRITE_SIZE*1024 (x2 = gfx950 FETCH_SIZE correction of the microarch guide)"}
json.dump(t, open(tpath, "w"), indent=1)
print(key, t[key]["hbm_bytes_per_launch"], "bytes per launch =", round((fetch + write) / (alg), 3), "x the algorithmic bytes of one step")
