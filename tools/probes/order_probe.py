import sys
import numpy as np
order = sys.argv[1]
if order == "torch_first":
    import torch
import os
from latticeboltzmannsimulations_amd import CavitySolver
if os.environ.get("LBM_PROBE_NO_ONE_RCCL"):          # reproduce the order solver._one_rccl() exists to prevent
    from latticeboltzmannsimulations_amd import solver as _solver
    _solver._one_rccl = lambda: None
with CavitySolver(256, 300, 100.0, rows=(100, 96), dtype=np.float32) as s:
    if len(sys.argv) > 2:
        s.comm_loopback()
        s.step(12)
if order == "torch_last":
    import torch
    if len(sys.argv) > 3:
        import torch.multiprocessing as mp
print("done", order, sys.argv[2:], flush=True)
