"""y-slab decomposition of the cavity lattice over the GPUs of one node.

No reference counterpart: the reference is single-GPU (MRT_GPU.py:29, cuda.Device(0)).
Row y is the slow index of every device plane, so a slab and its one-row halos are
contiguous.  Per step and per interface each side sends ONE row of three populations:
its first row of k = 2, 5, 6 (cy = +1, moving to smaller y) to the lower-rank neighbour,
its last row of k = 4, 7, 8 (cy = -1) to the higher-rank neighbour.

Two transports drive the same kernels:

* ``RcclSlab`` -- the production path: the exchange runs inside ``lbm_step`` in C
  (ncclSend/ncclRecv on a second HIP stream, overlapped with the interior rows);
  torch.distributed is used only to hand the ncclUniqueId to every rank.
* ``HaloDriver`` -- the same launch-unit protocol driven from outside (single steps: edges -> interior -> finish around a
  one-row halo; multi-step units: S complete rows per side, then lbm_step_unit); transports: torch.distributed P2P (gloo with host buffers, or nccl with device
  buffers) and an in-process copy between several slabs on ONE device (``LocalSlabs``).
  Used by the tests (including the CPU gloo tests with a stand-in stepper) and as a
  portable path.
"""
import numpy as np

LOW, HIGH = 0, 1


def partition_rows(ny, nslabs):
    """Balanced contiguous row ranges [(y0, ny_local)] -- earlier slabs get the remainder."""
    if nslabs < 1 or ny < 2 * nslabs:
        raise ValueError(f"cannot cut {ny} rows into {nslabs} slabs of >= 2 rows")
    base, rem = divmod(ny, nslabs)
    out, y0 = [], 0
    for r in range(nslabs):
        n = base + (1 if r < rem else 0)
        out.append((y0, n))
        y0 += n
    return out


def neighbours(rank, nslabs):
    """(lower-y neighbour or None, higher-y neighbour or None)"""
    return (rank - 1 if rank > 0 else None, rank + 1 if rank < nslabs - 1 else None)


def min_rows(ny, nslabs):
    """The smallest slab of partition_rows(ny, nslabs): pass it as `min_rows` to every CavitySolver of the decomposition, so
    that all of them derive the same launch plan (lbm_params.ny_local_min)."""
    return min(n for _, n in partition_rows(ny, nslabs))


def _next_unit(st, left):
    """Steps of the next launch unit (steppers without a multi-step path advance one step at a time)."""
    return st.next_unit(left) if hasattr(st, "next_unit") else 1


class _UnitDriver:
    """The launch-unit schedule of lbm_step(), driven from outside with any transport (subclasses provide exchange() -- the
    one-row halo of three planes -- and exchange_rows(S) -- S complete rows per side).  Same protocol as the in-library RCCL
    path: a single step needs a current one-row halo; a unit of S > 1 steps is preceded by ONE exchange of the S complete
    rows next to each interface and contains no communication (lbm_step_unit); every step() call ends with a one-row
    exchange so that get_fields() can return the populations of the slabs' first and last rows."""

    def _steppers(self):
        raise NotImplementedError

    def _plan(self, left, first=True):
        units = {_next_unit(st, left) for st in self._steppers()}
        if len(units) != 1:
            raise RuntimeError(f"the slabs disagree on the next launch unit ({sorted(units)}): create them with the same min_rows")
        return units.pop()

    def step(self, nsteps=1):
        left = int(nsteps)
        multi = self._nslabs() > 1
        first = True
        while left > 0:
            S = self._plan(left, first)
            first = False
            if S > 1:
                if multi:
                    self.exchange_rows(S)
                for st in self._steppers():
                    st.step_unit(S)
                self._thin = False
            else:
                if multi and not self._thin:
                    self.exchange()
                for st in self._steppers():
                    st.step_edges()
                    st.step_interior()
                    st.step_finish()
                self._thin = False
            left -= S
        if multi and not self._thin:
            self.exchange()
        return self


class LocalSlabs(_UnitDriver):
    """Several slabs held by ONE process (e.g. all on one device): the halo of slab i is
    copied straight from slab i +- 1.  `steppers` expose the externally driven primitives of
    CavitySolver (halo_elems / halo_export / halo_import / step_edges / step_interior /
    step_finish, and for multi-step units next_unit / halo_rows_elems / halo_export_rows / halo_import_rows / step_unit)."""

    def __init__(self, steppers):
        self.s = list(steppers)
        n = self.s[0].halo_elems()
        dt = self.s[0].dtype
        self._up = [np.empty(n, dtype=dt) for _ in self.s]    # what slab i sends to i-1
        self._down = [np.empty(n, dtype=dt) for _ in self.s]  # what slab i sends to i+1
        self._thin = False
        if len(self.s) > 1:
            self.exchange()   # slabs that were already stepped get current ghost rows; harmless otherwise

    def _steppers(self):
        return self.s

    def _nslabs(self):
        return len(self.s)

    def exchange(self):
        S = len(self.s)
        for i, st in enumerate(self.s):
            if i > 0:
                st.halo_export(LOW, self._up[i].ctypes.data)
            if i < S - 1:
                st.halo_export(HIGH, self._down[i].ctypes.data)
        for i, st in enumerate(self.s):
            if i > 0:
                st.halo_import(LOW, self._down[i - 1].ctypes.data)
            if i < S - 1:
                st.halo_import(HIGH, self._up[i + 1].ctypes.data)
        self._thin = True

    def exchange_rows(self, nrows):
        """The deep halo of a unit of `nrows` steps: the nrows complete rows next to every interface, both ways."""
        S = len(self.s)
        dt = self.s[0].dtype
        n = self.s[0].halo_rows_elems(nrows)
        up = [np.empty(n, dtype=dt) for _ in self.s]
        down = [np.empty(n, dtype=dt) for _ in self.s]
        for i, st in enumerate(self.s):
            if i > 0:
                st.halo_export_rows(LOW, nrows, up[i].ctypes.data)
            if i < S - 1:
                st.halo_export_rows(HIGH, nrows, down[i].ctypes.data)
        for i, st in enumerate(self.s):
            if i > 0:
                st.halo_import_rows(LOW, nrows, down[i - 1].ctypes.data)
            if i < S - 1:
                st.halo_import_rows(HIGH, nrows, up[i + 1].ctypes.data)


class HaloDriver(_UnitDriver):
    """One slab per process, halos moved with torch.distributed point-to-point ops.

    backend 'gloo': host staging buffers (works without a GPU for the stand-in stepper and
    with a GPU through lbm_halo_export/import to host memory); backend 'nccl': device
    buffers, RCCL moves them over xGMI."""

    def __init__(self, stepper, rank, world, device="cpu", group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.st, self.rank, self.world, self.group = stepper, rank, world, group
        self.device = device
        self.tdt = torch.float32 if np.dtype(stepper.dtype) == np.float32 else torch.float64
        n = stepper.halo_elems()
        mk = lambda: torch.empty(n, dtype=self.tdt, device=device)  # noqa: E731
        self.lo, self.hi = neighbours(rank, world)
        self.send = {LOW: mk(), HIGH: mk()}
        self.recv = {LOW: mk(), HIGH: mk()}
        self._rows = {}         # nrows -> (send, recv) buffers of the deep exchange, made once per unit length
        self._thin = False
        if world > 1:
            self._agree(self._signature(), "launch plan (steps per launch, frame width, kernel path): create every slab with the same "
                                           "parameters and min_rows")
            self.exchange()

    def _steppers(self):
        return [self.st]

    def _nslabs(self):
        return self.world

    def _signature(self):
        """What shapes the unit sequence besides (steps left, raw): the plan items of lbm_describe (a stand-in stepper without
        describe(): its steps per launch)."""
        import zlib
        if hasattr(self.st, "describe"):
            d = self.st.describe()
            d = d if isinstance(d, dict) else dict(kv.split("=", 1) for kv in str(d).split())
            key = "|".join(f"{k}={d.get(k)}" for k in ("kernel", "steps_per_launch", "frame", "stream", "deep_halo", "frame_fused", "lazy_lag"))
        else:
            key = f"units={_next_unit(self.st, 1 << 20)}"
        return zlib.crc32(key.encode())

    def _agree(self, value, what):
        t = self.torch.tensor([value, -value], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        if int(t[0]) != value or int(-t[1]) != value:
            raise RuntimeError(f"the ranks disagree on the {what}")

    def _plan(self, left, first=True):
        # The unit sequence of a call is a pure function of (launch plan, steps of the call, raw lattice or not): the plan was compared
        # once in __init__, so ONE all-reduce per step() call -- on the first unit, which carries the other two -- keeps the ranks'
        # send / receive sequences matched; nothing sits between the units of a call.
        S = _next_unit(self.st, left)
        if first and self.world > 1:
            self._agree((int(left) << 8) | S, "next launch unit: create every slab with the same min_rows and step them together")
        return S

    def _move(self, send, recv, export, imp):
        dist = self.dist
        ops = []
        for side, peer in ((LOW, self.lo), (HIGH, self.hi)):
            if peer is None:
                continue
            export(side, send[side].data_ptr())
            ops.append(dist.P2POp(dist.isend, send[side], peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, recv[side], peer, group=self.group))
        if ops:
            if send[LOW].is_cuda:
                self.torch.cuda.synchronize()
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if send[LOW].is_cuda:
                self.torch.cuda.synchronize()
        for side, peer in ((LOW, self.lo), (HIGH, self.hi)):
            if peer is not None:
                imp(side, recv[side].data_ptr())

    def exchange(self):
        self._move(self.send, self.recv, self.st.halo_export, self.st.halo_import)
        self._thin = True

    def exchange_rows(self, nrows):
        if nrows not in self._rows:
            n = self.st.halo_rows_elems(nrows)
            mk = lambda: self.torch.empty(n, dtype=self.tdt, device=self.device)  # noqa: E731
            self._rows[nrows] = ({LOW: mk(), HIGH: mk()}, {LOW: mk(), HIGH: mk()})
        send, recv = self._rows[nrows]
        self._move(send, recv, lambda side, ptr: self.st.halo_export_rows(side, nrows, ptr),
                   lambda side, ptr: self.st.halo_import_rows(side, nrows, ptr))


def global_mean_u(solver, world=1, group=None, device="cpu"):
    """mean(u) over the WHOLE lattice -- the quantity MRT_GPU.py:883-889 tests for convergence -- from slabs: every rank reduces its
    own rows on its GPU (lbm_mean_u, 8 bytes leave the device) and ONE all-reduce of a double sums the row-weighted means
    (the only collective the path may want, once per Pinterval, not per step).  `solver`: a CavitySolver holding a
    slab (or anything with mean_u(), ny_local and ny)."""
    part = float(solver.mean_u()) * int(solver.ny_local)
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([part], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        part = float(t[0])
    return part / int(solver.ny)


def attach_rccl(solver, rank, world, group=None):
    """Production path: give `solver` (a CavitySolver holding slab `rank`) an RCCL
    communicator so that lbm_step() exchanges halos itself.  torch.distributed (any backend)
    only broadcasts the 128-byte ncclUniqueId created on rank 0."""
    import torch.distributed as dist
    from .solver import comm_unique_id
    box = [comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    solver.comm_init(world, rank, box[0])
    return solver
