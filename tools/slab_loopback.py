#!/usr/bin/env python3
"""Step time of ONE middle slab that exchanges its deep halos with itself through RCCL (ncclSend / ncclRecv to its own rank on the
communication stream): what a rank of an N-GPU run does per step, measured on one GPU.

    python3 tools/slab_loopback.py nx rows_per_slab [dtype=f32] [arith=fast] [kernel=stream] [key=value tuning ...]

Prints the slab's step time with the exchange, the same slab with the exchange skipped is not possible outside a debug build,
so the comparison is the lone lattice of the same size (walls instead of neighbours, no exchange).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latticeboltzmannsimulations_amd import CavitySolver  # noqa: E402


def main():
    nx, rows = int(sys.argv[1]), int(sys.argv[2])
    dtype = np.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.float32
    arith = sys.argv[4] if len(sys.argv) > 4 else "fast"
    tune, kernel = {}, "auto"
    for kv in sys.argv[5:]:
        k, v = kv.split("=")
        if k == "kernel":
            kernel = v
        else:
            tune[k] = int(v) if k in ("tb_steps", "frame_seg") else v not in ("0", "false")
    steps = 400
    with CavitySolver(nx, 3 * rows, 1000.0, RT="MRT", dtype=dtype, rows=(rows, rows), arith=arith, tuning=tune, kernel=kernel) as s:
        s.comm_loopback()
        s.copy_bandwidth(1 << 30, 30)
        s.step(41); s.sync()
        ms = min(s.time_steps(steps) for _ in range(3)) / steps
        print("slab %d x %d %s %s loopback: %7.2f us/step %7.1f GLUPS  unit=%d  %s" % (nx, rows, sys.argv[3] if len(sys.argv) > 3 else "f32", arith,
              ms * 1e3, nx * rows / ms / 1e6, s.next_unit(1000), s.describe()), flush=True)
    with CavitySolver(nx, rows, 1000.0, RT="MRT", dtype=dtype, arith=arith, tuning=tune, kernel=kernel) as s:
        s.copy_bandwidth(1 << 30, 30)
        s.step(41); s.sync()
        ms = min(s.time_steps(steps) for _ in range(3)) / steps
        print("lone %d x %d: %7.2f us/step %7.1f GLUPS  unit=%d" % (nx, rows, ms * 1e3, nx * rows / ms / 1e6, s.next_unit(1000)), flush=True)


if __name__ == "__main__":
    main()
