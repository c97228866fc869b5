"""The driver's sequence (wake-up copies, 5 warm-up steps, 20 timed steps), then the same 20 steps again: event time and wall time."""
import time
import numpy as np
from latticeboltzmannsimulations_amd import CavitySolver
for rep in range(2):
    with CavitySolver(4096, 4096, 1000.0, RT="MRT", dtype=np.float32, arith="fast") as s:
        s.copy_bandwidth(1 << 30, 100)
        s.step(5); s.sync()
        for i in range(4):
            t0 = time.perf_counter()
            ev = s.time_steps(20)
            s.sync()
            dt = (time.perf_counter() - t0) * 1e3
            print("solver %d call %d: events %.1f us  wall %.1f us" % (rep, i, ev * 1e3, dt * 1e3), flush=True)
