import sys
import numpy as np
order = sys.argv[1]
if order == "torch_first":
    import torch
from latticeboltzmannsimulations_amd import CavitySolver
with CavitySolver(256, 300, 100.0, rows=(100, 96), dtype=np.float32) as s:
    if len(sys.argv) > 2:
        s.comm_loopback()
        s.step(12)
if order == "torch_last":
    import torch
    if len(sys.argv) > 3:
        import torch.multiprocessing as mp
print("done", order, sys.argv[2:], flush=True)
