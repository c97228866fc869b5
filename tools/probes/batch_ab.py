"""A/B of the batch path (the Reynolds sweep of MRT_GPU_datagen.py: B lattices of 384^2 per launch)."""
import sys
import numpy as np
from latticeboltzmannsimulations_amd import CavityBatch
B, n = 64, 384
cases = [(np.float32, "fast", {}), (np.float32, "strict", {}), (np.float64, "fast", {})]
for seg in (24, 32, 48, 64, 96):
    cases.append((np.float32, "fast", {"frame_fused_batch": True, "frame_seg": seg}))
    cases.append((np.float32, "fast", {"frame_seg": seg}))
cases += [(np.float32, "strict", {"frame_fused_batch": True, "frame_seg": 32}), (np.float32, "strict", {"frame_fused_batch": True, "frame_seg": 64}),
          (np.float64, "fast", {"frame_fused_batch": True, "frame_seg": 32}), (np.float64, "fast", {"frame_fused_batch": True, "frame_seg": 24}),
          (np.float64, "fast", {"frame_fused_batch": True, "frame_seg": 48})]
if len(sys.argv) > 1:
    B = int(sys.argv[1])
for dt, arith, tune in cases:
    with CavityBatch(n, n, [100.0 + 10 * i for i in range(B)], RT="MRT", dtype=dt, arith=arith, tuning=tune) as s:
        s.step(41); s.sync()
        ms = min(s.time_steps(400) for _ in range(3)) / 400
        print(f"{B} x {n}^2 {np.dtype(dt).name} {arith} {tune}: {ms * 1e3:.2f} us/step  {B * n * n / ms / 1e6:.1f} GLUPS aggregate", flush=True)
