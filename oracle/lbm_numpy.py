"""CPU oracle (NumPy) for the D2Q9 lid-driven-cavity step.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (latticeboltzmannsimulations_amd/) never does.

This is a restatement, made by reading the source, of the reference's per-step update:

  semantics="mrt_py"  : /root/reference/MRT.py:286-453   (NumPy CPU script, fp64, SRT)
  semantics="mrt_gpu" : /root/reference/MRT_GPU.py:336-699 (CUDA text: funRT SRT/TRT/MRT + funBC)

PARITY STATUS: **parity unpinned at bit level.**  MRT.py cannot be imported in the build
image (it needs numba and numexpr, MRT.py:9,13,21, both absent) and the reference holds no
tests, golden vectors or committed outputs for this path; MRT_GPU.py needs PyCUDA/CUDA.
The oracle is pinned (a) at physics level by the reference's own data file GhiaData.csv
(tests/golden/ghia.npz), (b) by algebraic identities of the operators, and (c) against a
second, independently written restatement in C (oracle/lbm_ref.c).  SURVEY.md Appendix C
lists values from a survey-session run of MRT.py that used stand-in modules for the two
missing imports; they are kept in tests as a *recorded cross-check with that provenance*,
not as a reference-generated fixture.

Layout follows the reference host arrays: fin[k, x, y], y fastest, y = 0 is the moving lid
(MRT.py:192-209); direction k moves a population from (x, y) to (x + cx[k], y - cy[k])
(MRT.py:404-414, MRT_GPU.py:412-413 "j-c because -ve y axis").
"""
import numpy as np

# --- a1: lattice constants (MRT.py:138-160) -----------------------------------------
CX = np.array([0, 1, 0, -1, 0, 1, -1, -1, 1])
CY = np.array([0, 0, 1, 0, -1, 1, 1, -1, -1])
RIGHT = [1, 5, 8]   # cx > 0   (MRT.py:157 RightStencil)
LEFT = [3, 6, 7]    # cx < 0   (MRT.py:155 LeftStencil)
TOP = [2, 5, 6]     # cy > 0   (MRT.py:158 TopStencil)
BOT = [4, 7, 8]     # cy < 0   (MRT.py:160 BotStencil)
CENTH = [0, 1, 3]   # cy == 0  (MRT.py:159 CentHStencil)

# MRT matrices (MRT.py:163-183, MRT_GPU.py:593-612)
M_GS = np.array([
    [1, 1, 1, 1, 1, 1, 1, 1, 1],
    [-4, -1, -1, -1, -1, 2, 2, 2, 2],
    [4, -2, -2, -2, -2, 1, 1, 1, 1],
    [0, 1, 0, -1, 0, 1, -1, -1, 1],
    [0, -2, 0, 2, 0, 1, -1, -1, 1],
    [0, 0, 1, 0, -1, 1, 1, -1, -1],
    [0, 0, -2, 0, 2, 1, 1, -1, -1],
    [0, 1, -1, 1, -1, 0, 0, 0, 0],
    [0, 0, 0, 0, 0, 1, -1, 1, -1]], dtype=np.float64)
M_GS_INV = np.array([
    [1.0/9, -1.0/9, 1.0/9, 0, 0, 0, 0, 0, 0],
    [1.0/9, -1.0/36, -1.0/18, 1.0/6, -1.0/6, 0, 0, 1.0/4, 0],
    [1.0/9, -1.0/36, -1.0/18, 0, 0, 1.0/6, -1.0/6, -1.0/4, 0],
    [1.0/9, -1.0/36, -1.0/18, -1.0/6, 1.0/6, 0, 0, 1.0/4, 0],
    [1.0/9, -1.0/36, -1.0/18, 0, 0, -1.0/6, 1.0/6, -1.0/4, 0],
    [1.0/9, 1.0/18, 1.0/36, 1.0/6, 1.0/12, 1.0/6, 1.0/12, 0, 1.0/4],
    [1.0/9, 1.0/18, 1.0/36, -1.0/6, -1.0/12, 1.0/6, 1.0/12, 0, -1.0/4],
    [1.0/9, 1.0/18, 1.0/36, -1.0/6, -1.0/12, -1.0/6, -1.0/12, 0, 1.0/4],
    [1.0/9, 1.0/18, 1.0/36, 1.0/6, 1.0/12, -1.0/6, -1.0/12, 0, -1.0/4]], dtype=np.float64)


def weights(dtype):
    """MRT.py:142-144 / MRT_GPU.py:363."""
    t = 1.0 / 36. * np.ones(9)
    t[1:5] = 1.0 / 9.0
    t[0] = 4.0 / 9.0
    return t.astype(dtype)


def relaxation(Re, ny_global, uLB=0.08, omega_eps=1.2, omega_q=1.2):
    """a2: MRT.py:53-75 / MRT_GPU.py:63-93.  Defaults for (omega_eps, omega_q) follow
    MRT_GPU.py:90; MRT.py:72 uses (1.0, 1.2)."""
    nuLB = uLB * ny_global / Re
    omega = 2.0 / (6. * nuLB + 1)
    delTRT = 1.0 / 3.5
    omegam = 1.0 / (0.5 + (delTRT / ((1 / omega) - 0.5)))
    return dict(omega=omega, omegam=omegam, omega_e=1.0, omega_eps=omega_eps, omega_q=omega_q)


def equ(rho, ux, uy, t, promote=False):
    """a3: MRT.py:213-231 (association per Python precedence); returns feq[9, ...].

    promote (the CUDA text, MRT_GPU.py:410,652: `rho_l*t_g[k]*(1. + 3.0*cu + 9*0.5*cu*cu - 3.0*0.5*usqr)`): `rho_l*t_g[k]` is a
    float product; the `double` literals make the bracket, and its product with that float, double operations (C's usual arithmetic
    conversions); the store rounds ONCE to the lattice type.  On an fp64 lattice this is the plain evaluation."""
    R = rho.dtype.type
    usqr = ux * ux + uy * uy
    feq = np.empty((9,) + rho.shape, dtype=rho.dtype)
    for k in range(9):
        cu = int(CX[k]) * ux + int(CY[k]) * uy
        if promote:
            D = np.float64
            cud, rt = cu.astype(D), (rho * t[k]).astype(D)
            feq[k] = (rt * (((D(1.) + D(3.0) * cud) + (D(4.5) * cud) * cud) - D(1.5) * usqr.astype(D))).astype(rho.dtype)
        else:
            feq[k] = (rho * t[k]) * (((R(1.) + R(3.0) * cu) + (R(4.5) * cu) * cu) - R(1.5) * usqr)
    return feq


class CavityOracle:
    """Whole-domain stepper on reference-layout arrays fin[9, X, Y]."""

    def __init__(self, nx, ny, Re, uLB=0.08, semantics="mrt_py", collision="SRT",
                 dtype=np.float64, omega_eps=None, omega_q=None, ny_global=None, turb=0, promote=False):
        """promote: the sub-expressions of MRT_GPU.py's CUDA text that carry `double` literals (lines 385, 410, 638-642, 652) are
        evaluated in double and rounded once to the lattice type -- what C's usual arithmetic conversions prescribe for that text;
        everything else stays an operation of the lattice type.  (nvcc's default FMA contraction is a further, unknowable, step.)"""
        assert semantics in ("mrt_py", "mrt_gpu") and collision in ("SRT", "TRT", "MRT")
        assert not (promote and semantics != "mrt_gpu"), "promotion is a property of MRT_GPU.py's CUDA text"
        self.promote = bool(promote)
        assert not (turb and semantics == "mrt_py"), "the Smagorinsky closure exists only in MRT_GPU.py"
        self.turb = int(turb)
        self.nx, self.ny, self.uLB = nx, ny, uLB
        self.sem, self.coll = semantics, collision
        self.dtype = np.dtype(dtype)
        R = self.R = self.dtype.type
        if omega_eps is None:
            omega_eps = 1.0 if semantics == "mrt_py" else 1.2   # MRT.py:72 vs MRT_GPU.py:90
        if omega_q is None:
            omega_q = 1.2
        self.relax = relaxation(Re, ny if ny_global is None else ny_global, uLB, omega_eps, omega_q)
        self.t = weights(dtype)
        self.MI = M_GS_INV.astype(dtype)
        rl = self.relax
        self.omega_vec = [R(0.0), R(rl["omega_e"]), R(rl["omega_eps"]), R(0.0), R(rl["omega_q"]),
                          R(0.0), R(rl["omega_q"]), R(rl["omega"]), R(rl["omega"])]
        # A.7 init (MRT.py:206,260-268)
        self.rho = np.ones((nx, ny), dtype=dtype)
        self.u = np.zeros((2, nx, ny), dtype=dtype)
        iv = np.zeros((2, nx, ny), dtype=dtype)
        iv[0, :, 0] = uLB
        self.fin = equ(self.rho, iv[0], iv[1], self.t, self.promote)
        self.feq = self.fin.copy()          # feq_g starts as a copy of fin (MRT_GPU.py:325)
        self.fpost = self.fin.copy()
        self.nsteps = 0

    # --- a4 + a5: macroscopic fields with wall overrides ------------------------------
    def macros(self, f):
        R = self.R
        X, Y = self.nx, self.ny
        rho = ((((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7]) + f[8])
        ux = (((((f[1] - f[3]) + f[5]) - f[6]) - f[7]) + f[8]) / rho      # MRT.py:320
        uy = (((((f[2] - f[4]) + f[5]) + f[6]) - f[7]) - f[8]) / rho      # MRT.py:321
        rho = rho.copy()
        rho[:, 0] = ((f[0][:, 0] + f[1][:, 0]) + f[3][:, 0]) + R(2.) * ((f[2][:, 0] + f[5][:, 0]) + f[6][:, 0])  # MRT.py:337
        ux[0, 1:] = 0; uy[0, 1:] = 0                                       # MRT.py:341
        ux[X - 1, 1:] = 0; uy[X - 1, 1:] = 0
        ux[:, Y - 1] = 0; uy[:, Y - 1] = 0
        ux[:, 0] = R(self.uLB); uy[:, 0] = 0                               # MRT.py:342
        return rho, ux, uy

    # --- Smagorinsky relaxation rate (MRT_GPU.py:368-387; the Van Driest lines 370-373 are dead
    #     code because Cs2 is overwritten at 374) ------------------------------------------
    def smagorinsky_omega(self, f, feq_prev, rho_prev):
        R = self.R
        tau0 = R(1.0) / R(self.relax["omega"])
        # product = c_x c_y f_k + product, k = 0..8: only the diagonals contribute
        p1 = -f[8] + (f[7] + (-f[6] + f[5]))
        p2 = -feq_prev[8] + (feq_prev[7] + (-feq_prev[6] + feq_prev[5]))
        q = p1 - p2
        if self.promote:
            # MRT_GPU.py:385: tau*tau (float * float), Cs2 = 0.025f and abs(Qmf) are floats; 18*1.4142 is a double constant and
            # makes the rest double: one rounding at `tau =`
            D = np.float64
            tau = (D(0.5) * (D(tau0) + np.sqrt(D(tau0 * tau0) + ((D(18 * 1.4142) * D(R(0.025))) * np.abs(q).astype(D))
                                               / rho_prev.astype(D)))).astype(self.dtype)
            return R(1.0) / tau
        tau = R(0.5) * (tau0 + np.sqrt(tau0 * tau0 + ((R(18 * 1.4142) * R(0.025)) * np.abs(q)) / rho_prev))
        return R(1.0) / tau

    # --- a6: collision operators -------------------------------------------------------
    def collide(self, f, rho, feq, w_nu=None):
        R = self.R
        rl = self.relax
        if w_nu is None:
            w_nu = R(rl["omega"])
        if self.coll == "SRT":                                             # MRT.py:396 / MRT_GPU.py:413
            return f - w_nu * (f - feq)
        if self.coll == "TRT":                                             # MRT_GPU.py:455-462,514-525
            op, om = w_nu, R(rl["omegam"])
            fp = np.empty_like(f); fm = np.empty_like(f)
            ep = np.empty_like(f); em = np.empty_like(f)
            for a, b in ((2, 4), (5, 7), (6, 8), (1, 3)):
                fp[a] = R(0.5) * (f[a] + f[b]); fp[b] = fp[a]
                fm[a] = R(0.5) * (f[a] - f[b]); fm[b] = -fm[a]
                ep[a] = R(0.5) * (feq[a] + feq[b]); ep[b] = ep[a]
                em[a] = R(0.5) * (feq[a] - feq[b]); em[b] = -em[a]
            fp[0] = f[0]; fm[0] = 0; ep[0] = feq[0]; em[0] = 0
            return (f - op * (fp - ep)) - om * (fm - em)
        # MRT: MRT_GPU.py:633-655
        m = [None] * 9
        for k in range(9):
            acc = None
            for j in range(9):
                c = M_GS[k, j]
                if c == 0:
                    continue
                term = R(c) * f[j]
                acc = term if acc is None else acc + term
            m[k] = acc
        jx, jy = m[3], m[5]
        meq = [None] * 9
        meq[0] = rho
        meq[1] = R(-2.0) * rho + R(3.0) * (jx * jx + jy * jy)
        meq[2] = (R(-3.0) * (jx * jx + jy * jy) + rho) + R(9.0) * (((jx * jx) * jy) * jy)
        meq[3] = m[3]
        meq[4] = -jx + R(3.0) * ((jx * jx) * jx)
        meq[5] = m[5]
        meq[6] = -jy + R(3.0) * ((jy * jy) * jy)
        meq[7] = jx * jx - jy * jy
        meq[8] = jx * jy
        if self.promote:
            # MRT_GPU.py:638-642: the literals -2.0, 3.0, 9.0 make the sums double; products of jx, jy among themselves stay float
            D, dt = np.float64, self.dtype
            s2, p4 = jx * jx + jy * jy, ((jx * jx) * jy) * jy
            meq[1] = (D(-2.0) * rho.astype(D) + D(3.0) * s2.astype(D)).astype(dt)
            meq[2] = ((D(-3.0) * s2.astype(D) + rho.astype(D)) + D(9.0) * p4.astype(D)).astype(dt)
            meq[4] = ((-jx).astype(D) + D(3.0) * ((jx * jx) * jx).astype(D)).astype(dt)
            meq[6] = ((-jy).astype(D) + D(3.0) * ((jy * jy) * jy).astype(D)).astype(dt)
        wv = list(self.omega_vec)
        wv[7] = wv[8] = w_nu
        for k in range(9):
            m[k] = m[k] - wv[k] * (m[k] - meq[k])
        out = np.empty_like(f)
        for k in range(9):
            acc = None
            for j in range(9):
                c = self.MI[k, j]
                if c == 0:
                    continue
                term = c * m[j]
                acc = term if acc is None else acc + term
            out[k] = acc
        return out

    # --- a7: streaming windows (inclusive dest ranges) ---------------------------------
    def window(self, k):
        X, Y = self.nx, self.ny
        cx, cy = int(CX[k]), int(CY[k])
        if self.sem == "mrt_py":                                           # MRT.py:404-414
            xr = (1, X - 2) if cx > 0 else ((0, X - 3) if cx < 0 else (0, X - 1))
            yr = (0, Y - 3) if cy > 0 else ((1, Y - 2) if cy < 0 else (0, Y - 1))
        else:                                                              # MRT_GPU.py:412
            xr = (max(0, cx), X - 1 + min(0, cx))
            yr = (max(0, -cy), Y - 1 + min(0, -cy))
        return xr, yr

    def stream(self, fin, fpost):
        """fin_k(x, y) <- fpost_k(x - cx, y + cy) inside the window; other slots keep fin."""
        for k in range(9):
            (x0, x1), (y0, y1) = self.window(k)
            cx, cy = int(CX[k]), int(CY[k])
            fin[k, x0:x1 + 1, y0:y1 + 1] = fpost[k, x0 - cx:x1 + 1 - cx, y0 + cy:y1 + 1 + cy]

    # --- a8: population boundary conditions ---------------------------------------------
    def wall_bc(self, fin, feq):
        X, Y = self.nx, self.ny
        if self.sem == "mrt_py":                                           # MRT.py:450-453
            fin[RIGHT, 0, :] = feq[RIGHT, 0, :]
            fin[LEFT, X - 1, :] = -feq[RIGHT, X - 1, :] + (feq[LEFT, X - 1, :] + fin[RIGHT, X - 1, :])
            fin[TOP, :, Y - 1] = -feq[BOT, :, Y - 1] + (feq[TOP, :, Y - 1] + fin[BOT, :, Y - 1])
            fin[BOT, :, 0] = -feq[TOP, :, 0] + (feq[BOT, :, 0] + fin[TOP, :, 0])
            return
        # MRT_GPU.py:674-692: x rule (if / else if), then y rule (if / else if)
        x = 0
        fin[1, x] = (feq[1, x] - feq[3, x]) + fin[3, x]
        fin[5, x] = (feq[5, x] - feq[7, x]) + fin[7, x]
        fin[8, x] = (feq[8, x] - feq[6, x]) + fin[6, x]
        x = X - 1
        fin[3, x] = (-feq[1, x] + feq[3, x]) + fin[1, x]
        fin[6, x] = (-feq[8, x] + feq[6, x]) + fin[8, x]
        fin[7, x] = (-feq[5, x] + feq[7, x]) + fin[5, x]
        y = Y - 1
        fin[2, :, y] = (-feq[4, :, y] + feq[2, :, y]) + fin[4, :, y]
        fin[5, :, y] = (-feq[7, :, y] + feq[5, :, y]) + fin[7, :, y]
        fin[6, :, y] = (-feq[8, :, y] + feq[6, :, y]) + fin[8, :, y]
        y = 0
        fin[4, :, y] = (-feq[2, :, y] + feq[4, :, y]) + fin[2, :, y]
        fin[7, :, y] = (-feq[5, :, y] + feq[7, :, y]) + fin[5, :, y]
        fin[8, :, y] = (-feq[6, :, y] + feq[8, :, y]) + fin[6, :, y]

    # --- a9: one iteration -------------------------------------------------------------
    def step(self, n=1):
        for _ in range(n):
            w_nu = self.smagorinsky_omega(self.fin, self.feq, self.rho) if self.turb else None
            rho, ux, uy = self.macros(self.fin)
            feq = equ(rho, ux, uy, self.t, self.promote)
            fpost = self.collide(self.fin, rho, feq, w_nu)
            self.stream(self.fin, fpost)
            self.wall_bc(self.fin, feq)
            self.rho = rho
            self.u = np.stack([ux, uy])
            self.feq, self.fpost = feq, fpost
            self.nsteps += 1
        return self

    def set_state(self, fin):
        """New populations; the Smagorinsky history (previous feq, rho) is reset to the
        equilibrium / density of the new state (same convention as lbm_set_state)."""
        self.fin = np.array(fin, dtype=self.dtype, copy=True)
        self.nsteps = 0
        rho, ux, uy = self.macros(self.fin)
        self.rho, self.feq = rho, equ(rho, ux, uy, self.t, self.promote)

    def peek_macros(self):
        """rho, u that the NEXT iteration would compute from the current fin."""
        rho, ux, uy = self.macros(self.fin)
        return rho, np.stack([ux, uy])
