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
* ``HaloDriver`` -- externally driven split step (edges -> interior -> finish, then export ->
  transport -> import of the rows just written); transports: torch.distributed P2P (gloo with host buffers, or nccl with device
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


class LocalSlabs:
    """Several slabs held by ONE process (e.g. all on one device): the halo of slab i is
    copied straight from slab i +- 1.  `steppers` expose the split-step primitives of
    CavitySolver (halo_elems / halo_export / halo_import / step_edges / step_interior /
    step_finish)."""

    def __init__(self, steppers):
        self.s = list(steppers)
        n = self.s[0].halo_elems()
        dt = self.s[0].dtype
        self._up = [np.empty(n, dtype=dt) for _ in self.s]    # what slab i sends to i-1
        self._down = [np.empty(n, dtype=dt) for _ in self.s]  # what slab i sends to i+1
        if len(self.s) > 1:
            self.exchange()   # slabs that were already stepped get current ghost rows; harmless otherwise

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

    def step(self, nsteps=1):
        """Each step is followed by the exchange of the rows it wrote, so that on return every
        slab's ghost rows are current (get_fields needs them for the populations)."""
        for _ in range(nsteps):
            for st in self.s:
                st.step_edges()
                st.step_interior()
                st.step_finish()
            if len(self.s) > 1:
                self.exchange()
        return self


class HaloDriver:
    """One slab per process, halos moved with torch.distributed point-to-point ops.

    backend 'gloo': host staging buffers (works without a GPU for the stand-in stepper and
    with a GPU through lbm_halo_export/import to host memory); backend 'nccl': device
    buffers, RCCL moves them over xGMI."""

    def __init__(self, stepper, rank, world, device="cpu", group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.st, self.rank, self.world, self.group = stepper, rank, world, group
        tdt = torch.float32 if np.dtype(stepper.dtype) == np.float32 else torch.float64
        n = stepper.halo_elems()
        mk = lambda: torch.empty(n, dtype=tdt, device=device)  # noqa: E731
        self.lo, self.hi = neighbours(rank, world)
        self.send = {LOW: mk(), HIGH: mk()}
        self.recv = {LOW: mk(), HIGH: mk()}
        if world > 1:
            self.exchange()

    def exchange(self):
        dist = self.dist
        ops = []
        for side, peer in ((LOW, self.lo), (HIGH, self.hi)):
            if peer is None:
                continue
            self.st.halo_export(side, self.send[side].data_ptr())
            ops.append(dist.P2POp(dist.isend, self.send[side], peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, self.recv[side], peer, group=self.group))
        if ops:
            if self.send[LOW].is_cuda:
                self.torch.cuda.synchronize()
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if self.send[LOW].is_cuda:
                self.torch.cuda.synchronize()
        for side, peer in ((LOW, self.lo), (HIGH, self.hi)):
            if peer is not None:
                self.st.halo_import(side, self.recv[side].data_ptr())

    def step(self, nsteps=1):
        for _ in range(nsteps):
            self.st.step_edges()
            self.st.step_interior()
            self.st.step_finish()
            if self.world > 1:
                self.exchange()
        return self


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
