"""CPU stand-in for one slab (TEST DOUBLE, built on the oracle's operators).

Exposes the externally driven interface of CavitySolver -- single steps (halo_elems / halo_export / halo_import /
step_edges / step_interior / step_finish) and multi-step launch units (next_unit / halo_rows_elems / halo_export_rows /
halo_import_rows / step_unit) -- so that the host-side slab logic of the product
(latticeboltzmannsimulations_amd/slab.py: partition, neighbours, which planes leave through
which side, the launch-unit schedule, torch.distributed transport) can be exercised without a GPU.
Like the HIP library it keeps POST-COLLISION populations as its state, so a halo is one row
of three post-collision planes, exported before the step that streams them; the deep halo of a unit of S steps is the S
complete rows next to the interface (here: post-collision populations + the equilibrium the wall rules need), from which
the unit recomputes the neighbour's rows -- S rows of them after step 0, one row fewer after every step.
"""
import ctypes

import numpy as np

from oracle import lbm_numpy as on

LOW, HIGH = 0, 1
PLANES_OUT = {LOW: (2, 5, 6), HIGH: (4, 7, 8)}   # cy = +1 leaves towards smaller y, cy = -1 towards larger y


class SlabStandIn:
    def __init__(self, nx, ny, Re, rows, semantics="mrt_gpu", collision="MRT", dtype=np.float64, uLB=0.08, unit=1, _state=None):
        """unit: steps per launch unit this stand-in reports through next_unit (1 = single steps only; the HIP library
        derives it from the lattice size)."""
        self.o = on.CavityOracle(nx, ny, Re, uLB=uLB, semantics=semantics, collision=collision, dtype=dtype)
        self.nx, self.NY = nx, ny
        self.y0, self.n = rows
        self.dtype = np.dtype(dtype)
        self.unit = int(unit)
        self.nsteps = 0
        self.u = np.zeros((2, nx, self.n), dtype=dtype)
        self.rho = np.ones((nx, self.n), dtype=dtype)
        self.fpost = np.zeros((9, nx, self.n + 2), dtype=dtype)     # rows -1 .. n (ghost rows at both ends)
        self._deep = {}                                             # side -> (fpost rows, feq rows) received for the next unit
        if _state is None:
            self.fin = self.o.fin[:, :, self.y0:self.y0 + self.n].copy()
            self._collide_stage()
        else:                                                       # (an extended band built by step_unit)
            self.fpost[:, :, 1:self.n + 1], self.feq, self.fin = _state
            self._next = (self.rho, self.u)

    # moments with GLOBAL wall overrides (MRT.py:337,341-342), equilibrium, collision
    def _collide_stage(self):
        o, f, R = self.o, self.fin, self.o.R
        X, Y = self.nx, self.NY
        gy = np.arange(self.y0, self.y0 + self.n)
        rho = ((((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7]) + f[8])
        with np.errstate(all="ignore"):                             # (rows of an extended band that hold no data yet)
            ux = (((((f[1] - f[3]) + f[5]) - f[6]) - f[7]) + f[8]) / rho
            uy = (((((f[2] - f[4]) + f[5]) + f[6]) - f[7]) - f[8]) / rho
        rho = rho.copy()
        lid = gy == 0
        rho[:, lid] = (((f[0] + f[1]) + f[3]) + R(2.) * ((f[2] + f[5]) + f[6]))[:, lid]
        nl = ~lid
        ux[0, nl] = 0; uy[0, nl] = 0; ux[X - 1, nl] = 0; uy[X - 1, nl] = 0
        bot = gy == Y - 1
        ux[:, bot] = 0; uy[:, bot] = 0
        ux[:, lid] = R(o.uLB); uy[:, lid] = 0
        self._next = (rho, np.stack([ux, uy]))
        with np.errstate(all="ignore"):
            self.feq = on.equ(rho, ux, uy, o.t)
            self.fpost[:, :, 1:self.n + 1] = o.collide(f, rho, self.feq)

    # what slab.global_mean_u needs of a CavitySolver: the mean of u over the slab's own rows (lbm_mean_u), its rows, the lattice height
    @property
    def ny_local(self):
        return self.n

    @property
    def ny(self):
        return self.NY

    def mean_u(self):
        return float(np.asarray(self.u, dtype=np.float64).mean())

    def halo_elems(self):
        return 3 * self.nx

    def _ct(self):
        return ctypes.c_double if self.dtype == np.float64 else ctypes.c_float

    def _buf(self, ptr, shape=None):
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(self._ct())), shape=shape or (3, self.nx))

    def halo_export(self, side, ptr):
        row = 1 if side == LOW else self.n
        b = self._buf(ptr)
        for j, k in enumerate(PLANES_OUT[side]):
            b[j] = self.fpost[k, :, row]

    def halo_import(self, side, ptr):
        row = 0 if side == LOW else self.n + 1
        b = self._buf(ptr)
        for j, k in enumerate(PLANES_OUT[1 - side]):
            self.fpost[k, :, row] = b[j]

    def step_edges(self):
        pass

    def step_interior(self):
        o = self.o
        gy0, gy1 = self.y0, self.y0 + self.n - 1
        for k in range(9):                                   # a7, windows in GLOBAL rows
            (x0, x1), (w0, w1) = o.window(k)
            a, b = max(w0, gy0), min(w1, gy1)
            if a > b:
                continue
            cx, cy = int(on.CX[k]), int(on.CY[k])
            la, lb = a - self.y0, b - self.y0
            self.fin[k, x0:x1 + 1, la:lb + 1] = self.fpost[k, x0 - cx:x1 + 1 - cx, la + cy + 1:lb + cy + 2]
        # a8 wall rules on the rows this slab owns: run the oracle's rule on a padded copy
        Y = self.NY
        full_f = np.zeros((9, self.nx, Y), dtype=self.dtype)
        full_e = np.ones((9, self.nx, Y), dtype=self.dtype)
        full_f[:, :, gy0:gy1 + 1] = self.fin
        full_e[:, :, gy0:gy1 + 1] = self.feq
        o.wall_bc(full_f, full_e)
        self.fin = full_f[:, :, gy0:gy1 + 1].copy()
        self.rho, self.u = self._next
        self._collide_stage()

    def step_finish(self):
        self.nsteps += 1

    # ---- multi-step launch units (deep halo) --------------------------------------------------------------
    def next_unit(self, left):
        if self.unit <= 1 or self.nsteps == 0 or left < 1:   # (like the library: the first step after an upload is a single step)
            return 1 if left >= 1 else 0
        return self.unit if left >= self.unit else (left if left >= 3 else 1)

    def halo_rows_elems(self, nrows):
        return 3 * 9 * self.nx * nrows

    def halo_export_rows(self, side, nrows, ptr):
        # post-collision populations, the equilibrium the wall rules use, and the populations themselves (slots outside
        # their streaming window keep their value) -- the HIP library's complete rows hold the same information
        b = self._buf(ptr, (3, 9, self.nx, nrows))
        rows = slice(0, nrows) if side == LOW else slice(self.n - nrows, self.n)
        b[0] = self.fpost[:, :, 1:self.n + 1][:, :, rows]
        b[1] = self.feq[:, :, rows]
        b[2] = self.fin[:, :, rows]

    def halo_import_rows(self, side, nrows, ptr):
        b = self._buf(ptr, (3, 9, self.nx, nrows))
        self._deep[side] = (b[0].copy(), b[1].copy(), b[2].copy())

    def step_unit(self, S):
        """S steps without communication: the band [own rows + S neighbour rows per interface] steps S times; its outermost
        rows lack their outer neighbours (zeros are streamed in), which spoils one more row per step -- after S steps exactly
        the received rows, never an own one."""
        lo = self._deep.pop(LOW, None)
        hi = self._deep.pop(HIGH, None)
        nlo = lo[0].shape[2] if lo else 0
        nhi = hi[0].shape[2] if hi else 0
        assert (self.y0 == 0 or nlo >= S) and (self.y0 + self.n == self.NY or nhi >= S), "deep halo missing for this unit"
        parts_f = ([lo[0]] if lo else []) + [self.fpost[:, :, 1:self.n + 1]] + ([hi[0]] if hi else [])
        parts_e = ([lo[1]] if lo else []) + [self.feq] + ([hi[1]] if hi else [])
        parts_i = ([lo[2]] if lo else []) + [self.fin] + ([hi[2]] if hi else [])
        band = SlabStandIn(self.nx, self.NY, 1.0, (self.y0 - nlo, self.n + nlo + nhi), dtype=self.dtype,
                           _state=(np.concatenate(parts_f, axis=2), np.concatenate(parts_e, axis=2), np.concatenate(parts_i, axis=2)))
        band.o = self.o
        for _ in range(S):
            band.step_interior()
        own = slice(nlo, nlo + self.n)
        self.fin = band.fin[:, :, own].copy()
        self.feq = band.feq[:, :, own].copy()
        self.fpost[:, :, 1:self.n + 1] = band.fpost[:, :, 1:band.n + 1][:, :, own]
        self.rho, self.u = band.rho[:, own].copy(), band.u[:, :, own].copy()
        self._next = (band._next[0][:, own].copy(), band._next[1][:, :, own].copy())
        self.nsteps += S
