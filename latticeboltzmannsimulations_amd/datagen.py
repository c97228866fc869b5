"""Batched Reynolds-number sweep -- the reference's MRT_GPU_datagen.py: many independent cavity
solves (default Re = 100, 110, ... 5090 at 384 x 384, SRT + Smagorinsky), each run to the reference's convergence
criterion, producing the four arrays its CNN scripts consume:

    feq_initial.npy [9, X, Y]   the common initial equilibrium      (MRT_GPU_datagen.py:899)
    f_final.npy     [n, 9, X, Y] converged populations per Re        (MRT_GPU_datagen.py:879-900)
    u_final.npy     [n, 2, X, Y] converged velocity per Re           (MRT_GPU_datagen.py:879-901)
    Re_range.npy    [n]                                              (MRT_GPU_datagen.py:902)

The reference solves them one after another (MRT_GPU_datagen.py:57); a 384^2 lattice keeps an MI355X busy for a few
microseconds per step, so here `concurrent` lattices form ONE batch (`CavityBatch`, lbm_params.batch): the same launches
advance all of them, each with its own relaxation rates (measured: 7.8 us per step for one 384^2 lattice, 1.6 us per
lattice and step in a batch of 64).  Consecutive Reynolds numbers share a batch (similar convergence times); a lattice that
meets the criterion is recorded at that iteration and merely keeps stepping until the rest of its batch is done.  With
`devices` the batches are dealt to several GPUs of the node, one host thread per device.  Every solve is bit-identical to
running it alone.
"""
import os
from concurrent.futures import ThreadPoolExecutor
from timeit import default_timer as timer

import numpy as np

from .solver import CavityBatch


def _solve_batch(idx, Re_range, xsize, ysize, RT, turb, uLB, maxIt, Pinterval, tolerance, device, dtype, say, out, arith, convergence="host"):
    """Runs the lattices Re_range[idx] in lock step; fills out = (f_final, u_final, its) rows idx.  The per-lattice logic is
    the reference's loop body (MRT_GPU_datagen.py:707-731,862-871): a check after iteration It = 0, Pinterval, 2 Pinterval, ...
    (i.e. after It + 1 steps), `count` consecutive-or-not hits of |mean(u) - mean(u_past)| / uLB < tolerance, stop at count > 5."""
    f_final, u_final, its = out
    with CavityBatch(xsize, ysize, [float(Re_range[i]) for i in idx], RT=RT, uLB=uLB, dtype=dtype, turb=turb, device=device, arith=arith) as b:
        feq_initial = b.get_fields(want_fin=True, out_dtype=np.float32)[2][0]      # fin = equ(1, InitVel) = feq_initial
        count = [0] * len(idx)
        past = [0.0] * len(idx)
        open_ = set(range(len(idx)))
        It = 0
        while open_:
            b.step(It + 1 - b.steps_done)
            # the check value: NumPy's float32 mean of the downloaded field (the reference's own definition), or the mean reduced on
            # the device in double -- B doubles cross PCIe instead of B fields; the fields are then fetched only when a lattice stops
            means = b.mean_u() if convergence == "device" else None
            u = None if convergence == "device" else b.get_fields(out_dtype=np.float32)[0]
            finished = []
            for j in sorted(open_):
                mean_u = float(means[j]) if convergence == "device" else float(np.mean(u[j]))
                say("current Re is " + str(Re_range[idx[j]]) + " and iteration is " + str(It))
                say("current mean u is " + str(mean_u / uLB))
                if abs(mean_u - past[j]) / uLB < tolerance:
                    count[j] += 1
                    if count[j] > 5:
                        say("breaking out of loop because of convergence")
                        finished.append(j)
                past[j] = mean_u
            last = It + Pinterval > maxIt - 1          # no further check: the rest finishes the loop like the reference
            if finished:
                if u is None:
                    u = b.get_fields(out_dtype=np.float32)[0]
                fin = b.get_fields(want_fin=True, out_dtype=np.float32)[2]
                for j in finished:
                    f_final[idx[j]], u_final[idx[j]], its[idx[j]] = fin[j], u[j], b.steps_done
                    open_.discard(j)
            if last and open_:
                say("max iterations reached. More needed for convergence.")
                if maxIt > b.steps_done:
                    b.step(maxIt - b.steps_done)
                u, _, fin = b.get_fields(want_fin=True, out_dtype=np.float32)
                for j in open_:
                    f_final[idx[j]], u_final[idx[j]], its[idx[j]] = fin[j], u[j], b.steps_done
                open_ = set()
            It += Pinterval
    return feq_initial


def generate(Re_range=None, xsize=32 * 12, ysize=32 * 12, RT="SRT", turb=1, uLB=0.08, maxIt=3000000, Pinterval=10000,
             tolerance=0.0000001, OutputFolder="./output", save=True, concurrent=64, devices=(0,), dtype=np.float32,
             quiet=False, arith="strict", convergence="host"):
    """Returns (feq_initial, f_final, u_final, Re_range, iterations_per_Re); writes the four .npy files when `save`."""
    say = (lambda *a: None) if quiet else print
    Re_range = np.arange(100, 5100, 10) if Re_range is None else np.asarray(Re_range)   # MRT_GPU_datagen.py:55
    n = len(Re_range)
    tstart = timer()
    out = (np.zeros((n, 9, xsize, ysize), dtype=np.float32), np.zeros((n, 2, xsize, ysize), dtype=np.float32),
           np.zeros(n, dtype=np.int64))
    chunks = [list(range(i, min(n, i + concurrent))) for i in range(0, n, concurrent)]

    def work(k):
        return _solve_batch(chunks[k], Re_range, xsize, ysize, RT, turb, uLB, maxIt, Pinterval, tolerance,
                            devices[k % len(devices)], dtype, say, out, arith, convergence)
    if len(devices) > 1 and len(chunks) > 1:
        with ThreadPoolExecutor(max_workers=len(devices)) as pool:      # lbm_step runs in C with the GIL released
            feq = list(pool.map(work, range(len(chunks))))
    else:
        feq = [work(k) for k in range(len(chunks))]
    feq_initial = feq[0] if feq else None
    f_final, u_final, its = out
    if save:
        if not os.path.isdir(OutputFolder):
            os.makedirs(OutputFolder, exist_ok=True)
        np.save(os.path.join(OutputFolder, "feq_initial.npy"), feq_initial)
        np.save(os.path.join(OutputFolder, "f_final.npy"), f_final)
        np.save(os.path.join(OutputFolder, "u_final.npy"), u_final)
        np.save(os.path.join(OutputFolder, "Re_range.npy"), Re_range)
    say("TOTAL time elapsed is ", timer() - tstart, "seconds")
    return feq_initial, f_final, u_final, Re_range, its


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Reynolds-number sweep (drop-in for MRT_GPU_datagen.py)")
    ap.add_argument("--Re", type=float, nargs=3, default=[100, 5100, 10], metavar=("START", "STOP", "STEP"))
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--concurrent", type=int, default=64, help="lattices per batch")
    ap.add_argument("--Pinterval", type=int, default=10000)
    ap.add_argument("--maxIt", type=int, default=3000000)
    ap.add_argument("--OutputFolder", default="./output")
    ap.add_argument("--arith", choices=["strict", "fast"], default="strict", help="fast: agrees with strict to rounding, ~1.3x faster")
    ap.add_argument("--convergence", choices=["host", "device"], default="host",
                    help="device: the convergence test on lbm_mean_u (reduced on the GPU) instead of the downloaded field")
    a = ap.parse_args(argv)
    generate(np.arange(*a.Re), xsize=a.size, ysize=a.size, concurrent=a.concurrent, Pinterval=a.Pinterval, maxIt=a.maxIt,
             OutputFolder=a.OutputFolder, arith=a.arith, convergence=a.convergence)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
