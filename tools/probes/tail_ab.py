"""Driver-shaped sequence (wake-up copies, 5 warm-up steps, 20 timed steps) with and without tail units on the tile kernel."""
import sys, time
import numpy as np
from latticeboltzmannsimulations_amd import CavitySolver
for arith in ("fast", "strict"):
    for tt in (True, False, True, False):
        with CavitySolver(4096, 4096, 1000.0, RT="MRT", dtype=np.float32, arith=arith, tuning=dict(tail_tiles=tt)) as s:
            s.copy_bandwidth(1 << 30, 100)
            s.step(5); s.sync()
            t0 = time.perf_counter(); ev = s.time_steps(20); s.sync(); dt = time.perf_counter() - t0
            ev2 = [s.time_steps(20) for _ in range(4)]
            print(f"{arith} tail_tiles={int(tt)}: first call events {ev * 1e3:.0f} us wall {dt * 1e6:.0f} us -> {20 * 4096 * 4096 / dt / 1e9:.1f} GLUPS; later calls {min(ev2) * 1e3:.0f} us "
                  f"({20 * 4096 * 4096 / min(ev2) / 1e6:.1f} GLUPS)", flush=True)
