"""Batched Reynolds-number sweep -- the reference's MRT_GPU_datagen.py: many independent cavity
solves (default Re = 100, 110, ... 5090 at 384 x 384, SRT + Smagorinsky), each run to the reference's convergence
criterion, producing the four arrays its CNN scripts consume:

    feq_initial.npy [9, X, Y]   the common initial equilibrium      (MRT_GPU_datagen.py:899)
    f_final.npy     [n, 9, X, Y] converged populations per Re        (MRT_GPU_datagen.py:879-900)
    u_final.npy     [n, 2, X, Y] converged velocity per Re           (MRT_GPU_datagen.py:879-901)
    Re_range.npy    [n]                                              (MRT_GPU_datagen.py:902)

The reference solves them one after another (MRT_GPU_datagen.py:57); a 384^2 lattice keeps an MI355X busy for a few
microseconds per step, so here `concurrent` independent lattices are in flight at once, each on its own pair of HIP
streams (and, with `devices`, spread over several GPUs of the node): their kernels overlap on the device.  Every solve is
bit-identical to running it alone.
"""
import os
from concurrent.futures import ThreadPoolExecutor
from timeit import default_timer as timer

import numpy as np

from .solver import CavitySolver


def generate(Re_range=None, xsize=32 * 12, ysize=32 * 12, RT="SRT", turb=1, uLB=0.08, maxIt=3000000, Pinterval=10000,
             tolerance=0.0000001, OutputFolder="./output", save=True, concurrent=16, devices=(0,), dtype=np.float32,
             quiet=False, host_threads=4):
    """Returns (feq_initial, f_final, u_final, Re_range, iterations_per_Re); writes the four .npy files when `save`."""
    say = (lambda *a: None) if quiet else print
    Re_range = np.arange(100, 5100, 10) if Re_range is None else np.asarray(Re_range)   # MRT_GPU_datagen.py:55
    n = len(Re_range)
    tstart = timer()
    f_final = np.zeros((n, 9, xsize, ysize), dtype=np.float32)
    u_final = np.zeros((n, 2, xsize, ysize), dtype=np.float32)
    its = np.zeros(n, dtype=np.int64)
    feq_initial = None
    pending = list(range(n))
    active = []          # [index, solver, count, u_past_mean, next_It]
    pool = ThreadPoolExecutor(max_workers=min(concurrent, host_threads)) if host_threads > 1 else None
    while pending or active:
        while pending and len(active) < concurrent:
            i = pending.pop(0)
            s = CavitySolver(xsize, ysize, float(Re_range[i]), RT=RT, uLB=uLB, dtype=dtype, turb=turb,
                             device=devices[i % len(devices)])
            if feq_initial is None:
                feq_initial = s.get_fields(want_fin=True, out_dtype=np.float32)[2]     # fin = equ(1, InitVel) = feq_initial
            active.append([i, s, 0, 0.0, 0])
        # enqueue, for every active lattice, the iterations up to its next check; then collect.  One host thread per
        # lattice: lbm_step loops over thousands of launches in C (ctypes releases the GIL), and a single thread would
        # finish enqueuing lattice A before starting on B, leaving nothing to overlap on the device.
        def advance(a):
            a[1].step(a[4] + 1 - a[1].steps_done)          # the check of iteration It happens after It + 1 steps
        if pool is not None and len(active) > 1:
            list(pool.map(advance, active))
        else:
            for a in active:
                advance(a)
        still = []
        for a in active:
            i, s, count, past, It = a
            u, _, fin = s.get_fields(want_fin=True, out_dtype=np.float32)
            mean_u = float(np.mean(u))
            say("current Re is " + str(Re_range[i]) + " and iteration is " + str(It))
            say("current mean u is " + str(mean_u / uLB))
            done = False
            if abs(mean_u - past) / uLB < tolerance:                      # MRT_GPU_datagen.py:727-731
                count += 1
                if count > 5:
                    say("breaking out of loop because of convergence")
                    done = True
            if not done and It + Pinterval > maxIt - 1:                   # no further check: finish the loop like the reference
                say("max iterations reached. More needed for convergence.")
                if maxIt > s.steps_done:
                    s.step(maxIt - s.steps_done)
                    u, _, fin = s.get_fields(want_fin=True, out_dtype=np.float32)
                done = True
            if done:
                f_final[i], u_final[i], its[i] = fin, u, s.steps_done
                s.close()
            else:
                still.append([i, s, count, mean_u, It + Pinterval])
        active = still
    if pool is not None:
        pool.shutdown()
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
    ap.add_argument("--concurrent", type=int, default=16)
    ap.add_argument("--Pinterval", type=int, default=10000)
    ap.add_argument("--maxIt", type=int, default=3000000)
    ap.add_argument("--OutputFolder", default="./output")
    a = ap.parse_args(argv)
    generate(np.arange(*a.Re), xsize=a.size, ysize=a.size, concurrent=a.concurrent, Pinterval=a.Pinterval, maxIt=a.maxIt,
             OutputFolder=a.OutputFolder)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
