"""ctypes loader for the C oracle (oracle/lbm_ref.c).  TEST INFRASTRUCTURE ONLY --
see the header of lbm_ref.c; never imported by the product package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblbmref.so")
SEM = {"mrt_py": 0, "mrt_gpu": 1}
COLL = {"SRT": 0, "TRT": 1, "MRT": 2}


def build(force=False):
    if os.environ.get("LBMREF_SO"):           # tests/test_oracle.py: the AddressSanitizer / UBSan build of the same source
        return os.environ["LBMREF_SO"]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "lbm_ref.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for suf, ct in (("f64", ctypes.c_double), ("f32", ctypes.c_float)):
            p = ctypes.POINTER(ct)
            f = getattr(_lib, "lbmref_step_" + suf)
            f.restype = ctypes.c_int
            f.argtypes = [p, p, p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                          ctypes.POINTER(ctypes.c_double), ctypes.c_double, ctypes.c_int, p, ctypes.c_int]
            h = getattr(_lib, "lbmref_history_" + suf)
            h.restype = ctypes.c_int
            h.argtypes = [p, p, p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
            g = getattr(_lib, "lbmref_init_" + suf)
            g.restype = ctypes.c_int
            g.argtypes = [p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
        _lib.lbmref_set_threads.argtypes = [ctypes.c_int]
        _lib.lbmref_max_threads.restype = ctypes.c_int
    return _lib


def set_threads(n):
    """OpenMP threads for the oracle loops (default 1; results are thread-count invariant)."""
    lib().lbmref_set_threads(int(n))


def max_threads():
    return int(lib().lbmref_max_threads())


def relax_vector(relax):
    return np.array([relax["omega"], relax["omegam"], relax["omega_e"], relax["omega_eps"], relax["omega_q"]],
                    dtype=np.float64)


class CavityOracleC:
    """Same interface as oracle.lbm_numpy.CavityOracle, backed by lbm_ref.c."""

    def __init__(self, nx, ny, Re, uLB=0.08, semantics="mrt_py", collision="SRT", dtype=np.float64,
                 omega_eps=None, omega_q=None, ny_global=None, turb=0, promote=False):
        """promote: evaluate the sub-expressions of MRT_GPU.py's CUDA text that carry `double` literals in double and round once to
        the lattice type (C's usual arithmetic conversions; lbm_ref.c, PROMOTE).  No effect on an fp64 lattice."""
        from .lbm_numpy import relaxation
        assert not (promote and semantics != "mrt_gpu"), "promotion is a property of MRT_GPU.py's CUDA text"
        self.promote = int(bool(promote))
        self.turb = int(turb)
        self.nx, self.ny, self.uLB = nx, ny, uLB
        self.sem, self.coll = semantics, collision
        self.dtype = np.dtype(dtype)
        if omega_eps is None:
            omega_eps = 1.0 if semantics == "mrt_py" else 1.2
        if omega_q is None:
            omega_q = 1.2
        self.relax = relaxation(Re, ny if ny_global is None else ny_global, uLB, omega_eps, omega_q)
        self._w = relax_vector(self.relax)
        self._suf = "f64" if self.dtype == np.float64 else "f32"
        self._ct = ctypes.c_double if self.dtype == np.float64 else ctypes.c_float
        self.fin = np.empty((9, nx, ny), dtype=self.dtype)
        getattr(lib(), "lbmref_init_" + self._suf)(self._p(self.fin), nx, ny, uLB, self.promote)
        self.rho = np.ones((nx, ny), dtype=self.dtype)
        self.u = np.zeros((2, nx, ny), dtype=self.dtype)
        self.feq = self.fin.copy() if self.turb else None       # feq_g starts as a copy of fin (MRT_GPU.py:325)
        self.nsteps = 0

    def _p(self, a):
        return a.ctypes.data_as(ctypes.POINTER(self._ct))

    def step(self, n=1):
        rc = getattr(lib(), "lbmref_step_" + self._suf)(
            self._p(self.fin), self._p(self.rho), self._p(self.u), self.nx, self.ny, int(n),
            SEM[self.sem], COLL[self.coll], self._w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), self.uLB,
            self.turb, self._p(self.feq) if self.turb else None, self.promote)
        if rc != 0:
            raise RuntimeError("lbmref_step failed")
        self.nsteps += n
        return self

    def set_state(self, fin):
        self.fin = np.ascontiguousarray(fin, dtype=self.dtype).copy()
        self.nsteps = 0
        if self.turb:   # Smagorinsky history := equilibrium / density of the uploaded state
            getattr(lib(), "lbmref_history_" + self._suf)(self._p(self.fin), self._p(self.rho), self._p(self.feq),
                                                          self.nx, self.ny, self.uLB, self.promote)
