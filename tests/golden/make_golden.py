#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes are
plain data and are what travels to the GPU box.

What is generated, and from what:

* ``vtr_4x3.npz`` / ``vtr_5x4_f32.npz`` -- bytes of the ``.vtr`` file written by the
  reference's own ``VTKWrapper.saveToVTK`` (/root/reference/VTKWrapper.py:6-10 ->
  vendored pure-Python pyevtk/hl.py:122-192) for small seeded inputs, together with
  those inputs.  VTKWrapper/pyevtk import unmodified here (numpy only).
* ``ghia.npz`` -- the Ghia et al. (1982) table parsed from the reference's data file
  GhiaData.csv with exactly the slicing of /root/reference/MRT.py:104-116.

What is deliberately NOT generated: outputs of MRT.py itself.  MRT.py imports ``numba``
and ``numexpr`` (MRT.py:9,13,21), neither of which is installed in this image, so it
raises ModuleNotFoundError on import.  Libraries the image lacks stay absent: this
script does not fabricate stand-in modules or execute an edited copy of the script.
The LBM step oracle (oracle/) is therefore pinned at physics level by ghia.npz only,
and its bit-level parity with MRT.py is "parity unpinned" (see oracle/README.md).
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def make_vtr(name, X, Y, dtype):
    sys.path.insert(0, REF)
    from VTKWrapper import saveToVTK  # noqa: E402  (reference's own writer)

    rng = np.random.default_rng(1982)
    ux = rng.standard_normal((X, Y, 1)).astype(dtype)
    uy = rng.standard_normal((X, Y, 1)).astype(dtype)
    uz = np.zeros((X, Y, 1), dtype=dtype)
    rho = (1.0 + 0.01 * rng.standard_normal((X, Y, 1))).astype(dtype)
    grid = (np.arange(0, X, dtype="float64"), np.arange(0, Y, dtype="float64"),
            np.arange(0, 1, dtype="float64"))  # MRT.py:91-94
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            saveToVTK((ux, uy, uz), rho, "ldc", "00007", grid)
            raw = open("ldc.00007.vtr", "rb").read()
        finally:
            os.chdir(cwd)
    np.savez(os.path.join(HERE, name), ux=ux, uy=uy, uz=uz, rho=rho,
             gx=grid[0], gy=grid[1], gz=grid[2],
             vtr=np.frombuffer(raw, dtype=np.uint8))
    print(name, len(raw), "bytes")


def make_ghia():
    csv = os.path.join(REF, "GhiaData.csv")
    G = np.genfromtxt(csv, delimiter=",")[6:23, 1:]        # MRT.py:104
    V = np.genfromtxt(csv, delimiter=",")[25:39, 2:9]      # MRT.py:105
    np.savez(os.path.join(HERE, "ghia.npz"),
             table=G, vortices=V,
             Y=G[:, 0], X=G[:, 9],                          # MRT.py:106-107
             Re=np.array([100, 400, 1000, 3200, 5000, 7500, 10000]),
             Ux=G[:, 1:8], Uy=G[:, 10:17])                  # MRT.py:109-111 (Re_dict columns)
    print("ghia.npz", G.shape, V.shape)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures are already committed")
    make_vtr("vtr_4x3.npz", 4, 3, "float64")
    make_vtr("vtr_5x4_f32.npz", 5, 4, "float32")
    make_ghia()
