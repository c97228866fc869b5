"""Does the device's clock state at the start of the driver's 20 timed steps depend on what woke it up?  (a) 40 ms of plain copies (what
bench.py does), (b) the same followed by 20 ms of arithmetic-bound kernels of ANOTHER lattice, then 5 warm-up steps and the 20 timed steps."""
import time
import numpy as np
from latticeboltzmannsimulations_amd import CavitySolver
for mode in ("copies", "copies+compute", "copies", "copies+compute"):
    with CavitySolver(4096, 4096, 1000.0, RT="MRT", dtype=np.float32, arith="fast") as s, \
            CavitySolver(2048, 2048, 1000.0, RT="MRT", dtype=np.float32, arith="strict") as other:
        other.step(9); other.sync()
        s.copy_bandwidth(1 << 30, 100)
        if mode == "copies+compute":
            other.step(1000); other.sync()          # ~20 ms of the tile kernel on another lattice
        s.step(5); s.sync()
        t0 = time.perf_counter(); ev = s.time_steps(20); s.sync(); dt = time.perf_counter() - t0
        print(f"{mode:16s}: 20 timed steps: events {ev * 1e3:.0f} us, wall {dt * 1e6:.0f} us -> {20 * 4096 * 4096 / dt / 1e9:.1f} GLUPS", flush=True)
