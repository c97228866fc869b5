"""Host side of the hot path: a thin ctypes wrapper over liblbm_hip.so that mirrors how the
reference script drives PyCUDA (MRT_GPU.py:23-30,309-328,701-760): create device state,
upload / initialise populations, advance N steps, read u / rho / fin back in the reference's
host layout ([k, x, y], y fastest, y = 0 the lid)."""
import ctypes

import numpy as np

from . import _lib as L

_DT = {np.dtype(np.float32): L.LBM_F32, np.dtype(np.float64): L.LBM_F64}
_COLL = {"SRT": L.LBM_SRT, "TRT": L.LBM_TRT, "MRT": L.LBM_MRT}
_SEM = {"mrt_py": L.LBM_SEM_MRT_PY, "mrt_gpu": L.LBM_SEM_MRT_GPU}
_KERNEL = {"auto": L.LBM_KERNEL_AUTO, "generic": L.LBM_KERNEL_GENERIC, "vec": L.LBM_KERNEL_VEC, "tb": L.LBM_KERNEL_TB,
           "push": L.LBM_KERNEL_PUSH, "stream": L.LBM_KERNEL_STREAM}
_ARITH = {"strict": L.LBM_ARITH_STRICT, "fast": L.LBM_ARITH_FAST}
_LAYOUT = {"auto": L.LBM_LAYOUT_AUTO, "planes": L.LBM_LAYOUT_PLANES, "rows": L.LBM_LAYOUT_ROWS}


def relaxation(Re, ysize, uLB=0.08, omega_eps=1.2, omega_q=1.2):
    """Relaxation parameters exactly as the reference script derives them on the host
    (MRT_GPU.py:63-93): nuLB from the WHOLE lattice height, omega, TRT omegam with
    Lambda = 1/3.5, MRT vector (omega_e = 1, omega_eps = omega_q = 1.2)."""
    nuLB = uLB * ysize / Re
    omega = 2.0 / (6. * nuLB + 1)
    delTRT = 1.0 / 3.5
    omegam = 1.0 / (0.5 + (delTRT / ((1 / omega) - 0.5)))
    return dict(omega=omega, omegam=omegam, omega_e=1.0, omega_eps=omega_eps, omega_q=omega_q)


def _tuning(t):
    """(tb_steps, frame_seg, flags) of lbm_params from a dict of A/B switches (see CavitySolver)."""
    t = dict(t or {})
    tb, seg, flags = int(t.pop("tb_steps", 0) or 0), int(t.pop("frame_seg", 0) or 0), 0
    off = {"deep_halo": L.LBM_FLAG_NO_DEEP_HALO, "frame_fused": L.LBM_FLAG_FRAME_UNFUSED, "frame_lds": L.LBM_FLAG_NO_FRAME_LDS,
           "comm_priority": L.LBM_FLAG_COMM_PRIORITY_OFF, "frame_wide": L.LBM_FLAG_FRAME_NARROW, "edge_first": L.LBM_FLAG_NO_EDGE_FIRST,
           "edge_reserve": L.LBM_FLAG_NO_EDGE_RESERVE, "xcd_bands": L.LBM_FLAG_NO_XCD_BANDS,
           "tail_tiles": L.LBM_FLAG_NO_TAIL_TILES}
    on = {"frame_fused_batch": L.LBM_FLAG_FRAME_FUSED_BATCH, "eager_lag": L.LBM_FLAG_EAGER_LAG, "stream_pairs": L.LBM_FLAG_STREAM_PAIRS}
    if "stream_walls" in t:     # (three states: True / False force it, absent = the library's choice per operator variant)
        flags |= L.LBM_FLAG_STREAM_WALLS if t.pop("stream_walls") else L.LBM_FLAG_NO_STREAM_WALLS
    for k, bit in off.items():
        if not t.pop(k, True):
            flags |= bit
    for k, bit in on.items():
        if t.pop(k, False):
            flags |= bit
    nt = t.pop("nt", None)
    if nt is not None:
        flags |= L.LBM_FLAG_NT_ON if nt else L.LBM_FLAG_NT_OFF
    fb = t.pop("frame_beside", None)
    if fb is not None:
        flags |= L.LBM_FLAG_FRAME_BESIDE_ON if fb else L.LBM_FLAG_FRAME_BESIDE_OFF
    if t:
        raise ValueError(f"unknown tuning switches {sorted(t)}")
    return tb, seg, flags


class CavitySolver:
    """One lattice (or one y-slab of it) resident on one MI355X.

    xsize, ysize : whole lattice (MRT_GPU.py:53-54)
    Re, uLB      : MRT_GPU.py:47,57
    RT           : 'SRT' | 'TRT' | 'MRT' (MRT_GPU.py:48)
    semantics    : 'mrt_gpu' (full streaming windows + NEBB on four walls, MRT_GPU.py:412,674-692)
                   or 'mrt_py' (the CPU script's windows and wall rules, MRT.py:404-453)
    dtype        : float32 (what MRT_GPU.py stores, MRT_GPU.py:207) or float64 (what MRT.py computes in)
    rows         : (y0, ny_local) when this object holds only a slab
    kernel       : 'auto' | 'generic' (one thread per cell) | 'vec' (16 B per access, MRT_GPU.py semantics) |
                   'tb' (three to five steps per launch through LDS; what 'auto' picks when it applies) |
                   'push' (the reference's two-launch push scheme, for A/B only)
    layout       : device arrays 'planes' [k][y][x], 'rows' [y][k][x], 'auto' (= rows)
    arith        : 'strict' (default; the reference's operation order, bit-identical to the CPU restatement the tests check against) or 'fast' (MRT operator
                   in factored form, about half the arithmetic, agrees to rounding)
    min_rows     : slabs: the smallest ny_local of ALL slabs of the decomposition (lbm_params.ny_local_min) -- the launch plan is
                   derived from it, so that neighbours run the same exchange protocol
    tuning       : A/B switches of the launch plan, none of which changes a result: tb_steps (2..5 steps per launch; 2..8 with kernel='stream'),
                   frame_seg, and the boolean flags deep_halo, frame_fused, frame_fused_batch, frame_lds, nt, comm_priority,
                   eager_lag, frame_beside, frame_wide, edge_first, edge_reserve, xcd_bands, tail_tiles (lbm_params.tb_steps / frame_seg / flags)
    """

    def __init__(self, xsize, ysize, Re, RT="MRT", uLB=0.08, semantics="mrt_gpu", dtype=np.float32, turb=0,
                 device=0, rows=None, kernel="auto", layout="auto", omega_eps=None, omega_q=None, batch=1, arith="strict",
                 min_rows=None, tuning=None):
        self._h = None
        self.batch = int(batch)
        self._lead = getattr(self, "_lead", ())      # leading axes of the host arrays: (B,) for a CavityBatch
        self.lib = L.lib()
        self.nx, self.ny = int(xsize), int(ysize)
        self.dtype = np.dtype(dtype)
        if self.dtype not in _DT:
            raise ValueError("dtype must be float32 or float64")
        if RT not in _COLL:
            raise ValueError("RT must be 'SRT', 'TRT' or 'MRT'")
        if semantics not in _SEM:
            raise ValueError("semantics must be 'mrt_gpu' or 'mrt_py'")
        if omega_eps is None:
            omega_eps = 1.0 if semantics == "mrt_py" else 1.2      # MRT.py:72 vs MRT_GPU.py:90
        if omega_q is None:
            omega_q = 1.2
        self.Re, self.RT, self.uLB, self.semantics = float(Re), RT, float(uLB), semantics
        self._omega_eps, self._omega_q = omega_eps, omega_q
        self.relax = relaxation(self.Re, self.ny, self.uLB, omega_eps, omega_q)
        self.y0, self.ny_local = (0, self.ny) if rows is None else (int(rows[0]), int(rows[1]))
        self.turb = int(turb)
        p = _params(self.nx, self.ny, self.y0, self.ny_local, self.dtype, RT, semantics, kernel, turb, device, layout, self.batch, arith,
                    min_rows, tuning, self.uLB, self.relax)
        err = ctypes.create_string_buffer(512)
        h = self.lib.lbm_create(ctypes.byref(p), err, len(err))
        if not h:
            raise RuntimeError("lbm_create: " + err.value.decode())
        self._h = ctypes.c_void_p(h)

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.lbm_last_error(self._h).decode()}")

    def close(self):
        if self._h is not None:
            self.lib.lbm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _host(self, a, shape, name):
        shape = self._lead + tuple(shape)
        a = np.asarray(a)
        if a.dtype not in _DT or a.shape != shape or not a.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{name} must be a C-contiguous float32/float64 array of shape {shape}")
        return a

    # -- state in -----------------------------------------------------------------------
    def init_equilibrium(self):
        self._check(self.lib.lbm_init_equilibrium(self._h), "lbm_init_equilibrium")

    def set_state(self, fin):
        fin = self._host(fin, (9, self.nx, self.ny), "fin")
        self._check(self.lib.lbm_set_state(self._h, fin.ctypes.data, _DT[fin.dtype]), "lbm_set_state")

    def set_relaxation(self, index=0, Re=None, **rates):
        """Relaxation rates of lattice `index` for all later steps: from a Reynolds number (as the reference derives them,
        MRT_GPU.py:63-93) and / or explicit omega, omegam, omega_e, omega_eps, omega_q."""
        r = dict(self.relax if Re is None else relaxation(float(Re), self.ny, self.uLB, self._omega_eps, self._omega_q))
        unknown = set(rates) - set(r)
        if unknown:
            raise ValueError(f"unknown relaxation rates {sorted(unknown)}")
        r.update(rates)
        self._check(self.lib.lbm_set_relaxation(self._h, int(index), r["omega"], r["omegam"], r["omega_e"], r["omega_eps"],
                                                r["omega_q"]), "lbm_set_relaxation")
        return r

    # -- time loop ----------------------------------------------------------------------
    def step(self, nsteps=1):
        self._check(self.lib.lbm_step(self._h, int(nsteps)), "lbm_step")
        return self

    def sync(self):
        self._check(self.lib.lbm_sync(self._h), "lbm_sync")

    def time_steps(self, nsteps):
        ms = ctypes.c_double(0.0)
        self._check(self.lib.lbm_time_steps(self._h, int(nsteps), ctypes.byref(ms)), "lbm_time_steps")
        return ms.value

    @property
    def steps_done(self):
        return int(self.lib.lbm_steps_done(self._h))

    def describe(self):
        """The launch plan as a dict (lbm_describe): kernel of the multi-step units, steps per launch, frame width, ..."""
        buf = ctypes.create_string_buffer(512)
        if self.lib.lbm_describe(self._h, buf, len(buf)) < 0:
            raise RuntimeError("lbm_describe failed")
        out = {}
        for kv in buf.value.decode().split():
            k, v = kv.split("=", 1)
            out[k] = int(v) if v.lstrip("-").isdigit() else v
        return out

    def next_unit(self, steps_left):
        """Time steps the next launch unit of step() advances when `steps_left` remain (1 = a single step)."""
        n = int(self.lib.lbm_next_unit(self._h, int(steps_left)))
        if n < 0:
            raise RuntimeError("lbm_next_unit failed")
        return n

    # -- state out ----------------------------------------------------------------------
    def get_fields(self, want_fin=False, out_dtype=None, u=None, rho=None, fin=None):
        """Returns (u[2,X,Y], rho[X,Y]) (and fin[9,X,Y]).  Arrays are whole-lattice shaped;
        a slab writes only its rows y0:y0+ny_local."""
        dt = self.dtype if out_dtype is None else np.dtype(out_dtype)
        lead = self._lead
        u = np.zeros(lead + (2, self.nx, self.ny), dtype=dt) if u is None else self._host(u, (2, self.nx, self.ny), "u")
        rho = np.zeros(lead + (self.nx, self.ny), dtype=dt) if rho is None else self._host(rho, (self.nx, self.ny), "rho")
        if want_fin and fin is None:
            fin = np.zeros(lead + (9, self.nx, self.ny), dtype=dt)
        if fin is not None:
            fin = self._host(fin, (9, self.nx, self.ny), "fin")
        self._check(self.lib.lbm_get_fields(self._h, u.ctypes.data, rho.ctypes.data,
                                            fin.ctypes.data if fin is not None else None, _DT[u.dtype]),
                    "lbm_get_fields")
        return (u, rho, fin) if fin is not None else (u, rho)

    def mean_u(self):
        """mean(u) of the fields get_fields() would return, reduced on the device in double (lbm_mean_u): one float per
        lattice crosses PCIe instead of the field.  A batch returns an array [B]."""
        out = (ctypes.c_double * self.batch)()
        self._check(self.lib.lbm_mean_u(self._h, out), "lbm_mean_u")
        return np.array(out[:]) if self._lead else float(out[0])

    def get_tau(self, out_dtype=None):
        """tau + tau_turbulent per cell of the last iteration (taus_g, MRT_GPU.py:387); 1 / omega everywhere when turb = 0."""
        dt = self.dtype if out_dtype is None else np.dtype(out_dtype)
        tau = np.zeros(self._lead + (self.nx, self.ny), dtype=dt)
        self._check(self.lib.lbm_get_tau(self._h, tau.ctypes.data, _DT[tau.dtype]), "lbm_get_tau")
        return tau

    # -- checkpoint / restart (the reference has neither) -------------------
    def save_checkpoint(self, path):
        """Write the populations and the run parameters to `path` (.npz).  Restarting from it continues bit-identically
        (the state of the scheme is `fin`); with turb = 1 the one-step Smagorinsky history restarts from the state."""
        u, rho, fin = self.get_fields(want_fin=True)
        path = _npz(path)
        np.savez(path, fin=fin, steps_done=self.steps_done, nx=self.nx, ny=self.ny, Re=self.Re, RT=self.RT, uLB=self.uLB,
                 semantics=self.semantics, dtype=self.dtype.name, rows=np.array([self.y0, self.ny_local]), turb=self.turb, u=u, rho=rho)
        return path

    def load_checkpoint(self, path, strict=True):
        """Upload the populations of a checkpoint written by save_checkpoint; returns the number of steps the checkpointed
        run had done.  The lattice size must match; with `strict` (default) so must dtype, semantics, collision operator,
        closure and Reynolds number -- a continuation is bit-identical only then.  The array is the whole lattice, so a slab
        may restart from a checkpoint of the undivided lattice (each context reads its own rows) but not from another slab's."""
        with np.load(_npz(path), allow_pickle=False) as z:
            if int(z["nx"]) != self.nx or int(z["ny"]) != self.ny:
                raise ValueError("checkpoint lattice size differs")
            rows = tuple(int(v) for v in z["rows"]) if "rows" in z else (0, self.ny)
            if rows != (0, self.ny) and rows != (self.y0, self.ny_local):
                raise ValueError(f"checkpoint holds rows {rows} only, this context needs {(self.y0, self.ny_local)}")
            if strict:
                have = dict(dtype=str(z["dtype"]), semantics=str(z["semantics"]), RT=str(z["RT"]), Re=float(z["Re"]), uLB=float(z["uLB"]),
                            turb=int(z["turb"]) if "turb" in z else self.turb)
                want = dict(dtype=self.dtype.name, semantics=self.semantics, RT=self.RT, Re=self.Re, uLB=self.uLB, turb=self.turb)
                diff = {k: (have[k], want[k]) for k in want if have[k] != want[k]}
                if diff:
                    raise ValueError(f"checkpoint was written by a different run (checkpoint, this solver): {diff}")
            self.set_state(np.ascontiguousarray(z["fin"]))
            return int(z["steps_done"])

    # -- slab exchange primitives ---------------------------------------------------------
    def halo_elems(self):
        return int(self.lib.lbm_halo_elems(self._h))

    def halo_export(self, side, ptr):
        self._check(self.lib.lbm_halo_export(self._h, int(side), ctypes.c_void_p(ptr)), "lbm_halo_export")

    def halo_import(self, side, ptr):
        self._check(self.lib.lbm_halo_import(self._h, int(side), ctypes.c_void_p(ptr)), "lbm_halo_import")

    def step_edges(self):
        self._check(self.lib.lbm_step_edges(self._h), "lbm_step_edges")

    def step_interior(self):
        self._check(self.lib.lbm_step_interior(self._h), "lbm_step_interior")

    def step_finish(self):
        self._check(self.lib.lbm_step_finish(self._h), "lbm_step_finish")

    # multi-step launch units between slabs: S complete rows per side before the unit, no communication inside it
    def halo_rows_elems(self, nrows):
        return int(self.lib.lbm_halo_rows_elems(self._h, int(nrows)))

    def halo_export_rows(self, side, nrows, ptr):
        self._check(self.lib.lbm_halo_export_rows(self._h, int(side), int(nrows), ctypes.c_void_p(ptr)), "lbm_halo_export_rows")

    def halo_import_rows(self, side, nrows, ptr):
        self._check(self.lib.lbm_halo_import_rows(self._h, int(side), int(nrows), ctypes.c_void_p(ptr)), "lbm_halo_import_rows")

    def step_unit(self, unit_steps):
        self._check(self.lib.lbm_step_unit(self._h, int(unit_steps)), "lbm_step_unit")

    def comm_init(self, nranks, rank, uid_bytes):
        _one_rccl()
        buf = ctypes.create_string_buffer(bytes(uid_bytes), 128)
        self._check(self.lib.lbm_comm_init(self._h, int(nranks), int(rank), buf), "lbm_comm_init")

    def comm_loopback(self):
        """Diagnostic: run lbm_step's RCCL exchange path on one GPU (the slab is its own periodic neighbour)."""
        _one_rccl()
        self._check(self.lib.lbm_comm_loopback(self._h), "lbm_comm_loopback")

    def fma_rate(self, ms=20.0):
        """About `ms` milliseconds of packed fp32 FMAs on every CU; achieved TFLOP/s (lbm_fma_rate)."""
        g = ctypes.c_double(0.0)
        self._check(self.lib.lbm_fma_rate(self._h, float(ms), ctypes.byref(g)), "lbm_fma_rate")
        return g.value

    def copy_bandwidth(self, nbytes=1 << 30, iters=10):
        g = ctypes.c_double(0.0)
        self._check(self.lib.lbm_copy_bandwidth(self._h, int(nbytes), int(iters), ctypes.byref(g)), "lbm_copy_bandwidth")
        return g.value


class CavityBatch(CavitySolver):
    """B independent cavities of the same size and scheme, one per Reynolds number, advanced by the same launches -- the
    sweep MRT_GPU_datagen.py:55-57 runs one lattice after the other.  Host arrays carry a leading [B] axis:
    fin[B,9,X,Y], u[B,2,X,Y], rho[B,X,Y].  Every lattice evolves exactly as it would alone."""

    def __init__(self, xsize, ysize, Re_list, **kw):
        self.Re_list = [float(r) for r in Re_list]
        if not self.Re_list:
            raise ValueError("Re_list is empty")
        for bad in ("rows", "batch"):
            if kw.get(bad) is not None:
                raise ValueError(f"CavityBatch does not take `{bad}`")
        self._lead = (len(self.Re_list),)
        super().__init__(xsize, ysize, self.Re_list[0], batch=len(self.Re_list), **kw)
        self.relax_list = [self.set_relaxation(i, Re=r) for i, r in enumerate(self.Re_list)]

    def save_checkpoint(self, path):
        raise NotImplementedError("checkpoint the lattices of a batch one by one through get_fields / set_state")

    load_checkpoint = save_checkpoint


def _params(nx, ny, y0, ny_local, dtype, RT, semantics, kernel, turb, device, layout, batch, arith, min_rows, tuning, uLB, relax):
    """lbm_params of include/lbm.h from the Python-level arguments."""
    p = L.lbm_params()
    p.struct_size = ctypes.sizeof(L.lbm_params)
    p.nx, p.ny, p.y0, p.ny_local = int(nx), int(ny), int(y0), int(ny_local)
    p.dtype, p.collision, p.semantics = _DT[np.dtype(dtype)], _COLL[RT], _SEM[semantics]
    p.kernel, p.turb, p.device = _KERNEL[kernel], int(turb), int(device)
    p.layout = _LAYOUT[layout]
    p.batch = int(batch)
    p.arith = _ARITH[arith]
    p.ny_local_min = 0 if min_rows is None else int(min_rows)
    p.tb_steps, p.frame_seg, p.flags = _tuning(tuning)
    p.uLB = float(uLB)
    p.omega, p.omegam = relax["omega"], relax["omegam"]
    p.omega_e, p.omega_eps, p.omega_q = relax["omega_e"], relax["omega_eps"], relax["omega_q"]
    return p


def launch_plan(xsize, ysize, Re, steps=0, ncu=0, RT="MRT", uLB=0.08, semantics="mrt_gpu", dtype=np.float32, turb=0, rows=None,
                kernel="auto", layout="auto", batch=1, arith="strict", min_rows=None, tuning=None):
    """Dry run (lbm_plan, NO GPU needed): the launch plan lbm_create would derive for these arguments -- the dict of
    CavitySolver.describe() -- plus `units`, the launch units step(steps) would run from a fresh lattice.  All ranks of a slab
    decomposition must agree on kernel / steps_per_launch / frame / deep_halo and on the units: a launcher (bench.py --gpus N) and
    the CPU tests check that before any rank touches a device."""
    ny = int(ysize)
    y0, nyl = (0, ny) if rows is None else (int(rows[0]), int(rows[1]))
    relax = relaxation(float(Re), ny, float(uLB), 1.0 if semantics == "mrt_py" else 1.2, 1.2)
    p = _params(xsize, ny, y0, nyl, dtype, RT, semantics, kernel, turb, 0, layout, batch, arith, min_rows, tuning, uLB, relax)
    buf = ctypes.create_string_buffer(1024)
    rc = L.lib().lbm_plan(ctypes.byref(p), int(ncu), int(steps), buf, len(buf))
    text = buf.value.decode()
    if rc < 0:
        raise RuntimeError("lbm_plan: " + text)
    out = {}
    for kv in text.split():
        k, v = kv.split("=", 1)
        out[k] = int(v) if v.lstrip("-").isdigit() else v
    out["units"] = [int(x) for x in str(out.get("units", "")).split(",") if x != ""]
    return out


def _npz(path):
    """np.savez appends '.npz' to a name without it; use one spelling for writing and reading."""
    path = str(path)
    return path if path.endswith(".npz") else path + ".npz"


def _one_rccl():
    """PyTorch before RCCL -- belt and braces since r03.  r02: a process that created a communicator through the library and imported
    torch only AFTERWARDS aborted in exit() ("double free or corruption").  r03 found the cause and fixed it in the library
    (lbm_comm.hip, rccl()): RCCL had been opened RTLD_GLOBAL, which put its dependency librocm_smi64 into the global symbol scope, and a
    library mapped later that defines the same namespace-scope std::map<amd::smi::DevInfoTypes, const char*> (the wheel's librocm_smi64,
    /opt/rocm's libamd_smi.so) bound its initialiser and destructor to that one object: destroyed twice (backtraces
    profiles/r03_logs/rc134_gdb.log, rc134_gdb2.log; after the fix every order exits 0: rccl_order_r03.log).  RCCL is now opened
    RTLD_LOCAL and taken from the directory of the HIP runtime in use.  Importing torch first, where it is installed, costs nothing:
    processes that use this path exchange the communicator id through torch.distributed anyway."""
    import importlib.util
    import sys
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def comm_unique_id():
    _one_rccl()
    buf = ctypes.create_string_buffer(128)
    if L.lib().lbm_comm_unique_id(buf) != 0:
        raise RuntimeError("lbm_comm_unique_id failed")
    return buf.raw
