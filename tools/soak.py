#!/usr/bin/env python3
"""Long runs: the multi-step launch paths against one step per launch, bit for bit, and run to run.

    python3 tools/soak.py lone      4096² / 1024² lone lattices: auto (streaming / tile kernel, irregular call lengths) twice, and kernel = vec
    python3 tools/soak.py slab      one middle slab in RCCL loopback: units under the streaming kernel (edge + bulk launch, exchange under
                                    the bulk launch) against one step and one exchange per launch
"""
import hashlib
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latticeboltzmannsimulations_amd import CavitySolver  # noqa: E402


def lone():
    def final(n, dt, arith, kernel, steps, chunks=7):
        with CavitySolver(n, n, 1000.0, RT="MRT", dtype=dt, arith=arith, kernel=kernel) as s:
            t = time.time()
            per = steps // chunks
            for i in range(chunks):
                s.step(per + (i % 3))          # irregular call lengths: different tails
            tot = s.steps_done
            s.sync(); dtm = time.time() - t
            u, rho, fin = s.get_fields(want_fin=True)
            return tot, fin, u, dtm, s.describe()["kernel"]
    for n, dt, arith, steps in ((4096, np.float32, "fast", 100000), (4096, np.float32, "strict", 30000), (4096, np.float64, "fast", 30000),
                                (2048, np.float32, "fast", 100000)):
        t1, f1, u1, d1, k1 = final(n, dt, arith, "auto", steps)
        t2, f2, u2, d2, _ = final(n, dt, arith, "auto", steps)
        assert t1 == t2
        with CavitySolver(n, n, 1000.0, RT="MRT", dtype=dt, arith=arith, kernel="vec") as s:   # one step per launch, one call
            s.step(t1); s.sync()
            u3, rho3, f3 = s.get_fields(want_fin=True)
        print(f"{n}^2 {np.dtype(dt).name} {arith} ({k1}): {t1} steps in {d1:.1f} s; run-to-run identical: {np.array_equal(f1, f2)}; "
              f"multi-step == one-step kernel: {np.array_equal(f1, f3) and np.array_equal(u1, u3)}; finite: {bool(np.isfinite(f1).all())}; "
              f"max |u|/uLB {np.abs(u1).max() / 0.08:.3f}", flush=True)


def slab():
    for nx, rows, dt, arith, calls in ((4096, 1024, np.float32, "fast", (3001, 2999, 4003)), (4096, 512, np.float32, "strict", (3001, 2999, 4003)),
                                       (16384, 2048, np.float32, "fast", (1001, 999, 1003)), (8192, 1024, np.float64, "fast", (1001, 999, 1003))):
        digests = []
        for kernel in ("auto", "auto", "vec"):
            with CavitySolver(nx, 3 * rows, 1000.0, RT="MRT", dtype=dt, rows=(rows, rows), arith=arith, kernel=kernel) as s:
                s.comm_loopback()
                s.set_state(smooth_state(nx, 3 * rows, dt))
                t = time.time()
                for c in calls:
                    s.step(c)
                s.sync(); dtm = time.time() - t
                u, rho, fin = s.get_fields(want_fin=True)
                own = fin[:, :, rows:2 * rows]
                digests.append((hashlib.sha256(own.tobytes()).hexdigest()[:16], hashlib.sha256(u[:, :, rows:2 * rows].tobytes()).hexdigest()[:16]))
                print(f"  slab {nx} x {rows} {np.dtype(dt).name} {arith} kernel={kernel} ({s.describe()['kernel']}, unit {s.next_unit(100)}): {s.steps_done} steps "
                      f"in {dtm:.1f} s  fin {digests[-1][0]}  u {digests[-1][1]}  finite {bool(np.isfinite(own).all())}", flush=True)
        print(f"slab {nx} x {rows} {np.dtype(dt).name} {arith}: run-to-run identical: {digests[0] == digests[1]}; units == one step and one exchange per launch: "
              f"{digests[0] == digests[2]}", flush=True)




def smooth_state(nx, ny, dt):
    """A smooth non-trivial state (a slab in loopback never sees the lid: from the rest state every row stays equal to every other,
    and a kernel that read a stale or a not-yet-written row would go unnoticed)."""
    x = np.arange(nx, dtype=np.float64)[:, None]
    y = np.arange(ny, dtype=np.float64)[None, :]
    base = 1.0 + 1e-3 * np.sin(0.01 * x) * np.cos(0.013 * y) + 5e-4 * np.cos(0.0037 * (x + 2 * y))
    t = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)
    fin = np.empty((9, nx, ny), dtype=dt)
    for k in range(9):
        fin[k] = (base * (t[k] * (1.0 + 1e-4 * k))).astype(dt)
    return fin


def one(argv):
    """python3 tools/soak.py one nx rows f32|f64 arith kernel [key=value ...] [calls=a,b,c] -> digest of the slab's populations"""
    nx, rows, dt, arith, kernel = int(argv[0]), int(argv[1]), np.float64 if argv[2] == "f64" else np.float32, argv[3], argv[4]
    tune, calls = {}, (3001, 2999, 4003)
    for kv in argv[5:]:
        k, v = kv.split("=")
        if k == "calls":
            calls = tuple(int(x) for x in v.split(","))
        else:
            tune[k] = int(v) if k in ("tb_steps", "frame_seg") else v not in ("0", "false")
    with CavitySolver(nx, 3 * rows, 1000.0, RT="MRT", dtype=dt, rows=(rows, rows), arith=arith, kernel=kernel, tuning=tune) as s:
        s.comm_loopback()
        s.set_state(smooth_state(nx, 3 * rows, dt))
        for c in calls:
            s.step(c)
        u, rho, fin = s.get_fields(want_fin=True)
        own = fin[:, :, rows:2 * rows]
        print(f"{' '.join(argv):70s} {s.describe()['kernel']:13s} fin {hashlib.sha256(own.tobytes()).hexdigest()[:12]}  u {hashlib.sha256(u[:, :, rows:2 * rows].tobytes()).hexdigest()[:12]}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "one":
        one(sys.argv[2:])
    elif sys.argv[1] == "seq":      # the same slab several times in ONE process: seq <n> <args of `one`>
        for _ in range(int(sys.argv[2])):
            one(sys.argv[3:])
    else:
        {"lone": lone, "slab": slab}[sys.argv[1]]()
