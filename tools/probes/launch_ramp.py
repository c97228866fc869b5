"""How long is each 8-step launch of the headline lattice right after the driver-shaped start (wake-up, 5 warm-up steps)?  Prints the
HIP-event time of successive single launches: a ramp here is what separates the driver's 20-step figure from the long-run one.

    python tools/probes/launch_ramp.py fast|strict [none|copies|copies+fma|fma|workN (N steps of the workload itself)] [reps]

Run one process per wake-up mode, with a few seconds of idle device between them (the first lattice of a process is the cold one)."""
import sys
import numpy as np
from latticeboltzmannsimulations_amd import CavitySolver
arith = sys.argv[1] if len(sys.argv) > 1 else "fast"
wake = sys.argv[2] if len(sys.argv) > 2 else "copies+fma"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for rep in range(reps):
    with CavitySolver(4096, 4096, 1000.0, RT="MRT", dtype=np.float32, arith=arith) as s:
        if "copies" in wake:
            s.copy_bandwidth(1 << 30, 100)
        if "fma" in wake:
            s.fma_rate(20.0)
        if "work" in wake:                       # ~25 ms of the workload's own kernel, then back to the initial state
            s.step(int(wake.split("work")[1] or 600)); s.sync()
            s.init_equilibrium()
        s.step(5); s.sync()
        S = s.next_unit(1000)
        t = [s.time_steps(S) * 1e3 for _ in range(40)]
        print(f"{arith} wake={wake} rep {rep} S={S} us per launch:", " ".join(f"{v:.0f}" for v in t), flush=True)
        t = [s.time_steps(20) * 1e3 for _ in range(4)]
        print("   20-step calls:", " ".join(f"{v:.0f}" for v in t), flush=True)
