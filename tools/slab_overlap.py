#!/usr/bin/env python3
"""Evidence for the overlap claim of the slab path: one middle slab in RCCL loopback (it exchanges deep halos
with itself through ncclSend / ncclRecv on the communication stream) stepped through multi-step launch units.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/slab -o slab -- python3 tools/slab_overlap.py run [nx rows]
    python3 tools/slab_overlap.py summarize gpurun_out/prof/slab profiles/r02_slab_loopback_overlap.csv

`summarize` reads the kernel trace and writes, per launch unit, the interval of the tile kernel (compute stream) and of the frame
kernel and the RCCL kernel (communication stream), and how much of the latter two lies inside the tile kernel's interval.
"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    from latticeboltzmannsimulations_amd import CavitySolver
    nx = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    nr = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    NY, rows = 3 * nr, (nr, nr)
    s = CavitySolver(nx, NY, 1000.0, RT="MRT", dtype=np.float32, rows=rows, arith="fast")
    s.comm_loopback()
    s.copy_bandwidth(1 << 30, 30)
    s.step(41); s.sync()
    ms = s.time_steps(200)
    print("loopback slab %dx%d fp32 fast: %.2f us per step, plan %s" % (nx, nr, ms / 200 * 1e3, s.describe()))
    s.close()


def summarize(src, dst):
    path = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "")) for r in rows), key=lambda t: t[0])
    # the bulk launch of a unit: the tile kernel, or the streaming kernel's launch on the compute stream (its edge launch -- interface
    # rows + column strips -- goes to the communication stream, the one the RCCL kernels run on, and counts as the unit's frame work)
    is_rccl = lambda name: "rcclgeneric" in name.lower() or "nccldevkernel" in name.lower() or "msccl" in name.lower()  # noqa: E731
    comm = {e[3] for e in ev if is_rccl(e[2])}
    tiles = [e for e in ev if "k_stepS_deep" in e[2] or ("k_stream" in e[2] and e[3] not in comm)]
    frames = [e for e in ev if "k_frame_multi" in e[2] or "k_step_frame" in e[2] or ("k_stream" in e[2] and e[3] in comm)]
    rccl = [e for e in ev if is_rccl(e[2])]
    t0 = tiles[0][0]
    def inside(e, t):
        return max(0, min(e[1], t[1]) - max(e[0], t[0]))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        f.write("# tools/slab_overlap.py: one fp32 slab in RCCL loopback, times in us from the first bulk kernel (tile = the bulk launch of a unit:\n")
        f.write("# k_stepS_deep, or k_stream's large-grid launch; frame = k_frame_multi, or k_stream's edge launch); *_in_tile = part of that\n")
        f.write("# kernel's run time that lies inside the interval of a bulk kernel (compute stream)\n")
        w.writerow(["unit", "tile_start", "tile_end", "tile_us", "frame_us", "frame_in_tile_us", "rccl_us", "rccl_in_tile_us"])
        tot = [0.0, 0.0, 0.0, 0.0, 0.0]
        for i, t in enumerate(tiles):
            nxt = tiles[i + 1][0] if i + 1 < len(tiles) else t[1]
            fr = [e for e in frames if t[0] - 1000_000 < e[0] < nxt and e[0] >= (tiles[i - 1][0] if i else 0)]
            rc = [e for e in rccl if e[0] >= (tiles[i - 1][1] if i else 0) - 0 and e[0] < nxt]
            fr = [e for e in fr if e[0] >= t[0] - 200_000]
            f_us = sum(e[1] - e[0] for e in fr) / 1e3; f_in = sum(max(inside(e, tt) for tt in tiles) for e in fr) / 1e3
            rc = [e for e in rc if e[0] >= t[0] - 200_000]
            r_us = sum(e[1] - e[0] for e in rc) / 1e3; r_in = sum(max(inside(e, tt) for tt in tiles) for e in rc) / 1e3
            if 20 <= i < 40:
                w.writerow([i, round((t[0] - t0) / 1e3, 1), round((t[1] - t0) / 1e3, 1), round((t[1] - t[0]) / 1e3, 1), round(f_us, 1), round(f_in, 1),
                            round(r_us, 1), round(r_in, 1)])
            for j, v in enumerate(((t[1] - t[0]) / 1e3, f_us, f_in, r_us, r_in)):
                tot[j] += v
        n = len(tiles)
        w.writerow(["mean_of_%d" % n] + ["", ""] + [round(v / n, 1) for v in tot])
        span = (tiles[-1][1] - tiles[0][0]) / 1e3
        w.writerow(["span_us", round(span, 1), "tile_busy_frac", round(tot[0] / span, 3), "", "", "", ""])
    print(open(dst).read())


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarize(sys.argv[2], sys.argv[3])
