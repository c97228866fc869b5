"""Multi-slab host logic on CPU: (1) several slabs in one process, (2) world_size-2 and -3
torch.distributed runs over gloo, incl. world 8 with multi-step launch units.  The compute is the stand-in stepper (tests/slab_standin.py);
the code under test is the product's slab.py (partition, neighbours, side/plane conventions,
driver ordering, P2P transport).  The result must be bit-identical to the undivided oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import lbm_numpy as on                                   # noqa: E402
from latticeboltzmannsimulations_amd.slab import LocalSlabs, partition_rows  # noqa: E402
from slab_standin import SlabStandIn                                  # noqa: E402


@pytest.mark.parametrize("sem,coll", [("mrt_gpu", "MRT"), ("mrt_py", "SRT"), ("mrt_gpu", "TRT")])
@pytest.mark.parametrize("nslabs", [2, 3, 5])
def test_local_slabs_equal_single_domain(sem, coll, nslabs):
    nx, ny, steps = 20, 23, 12
    ref = on.CavityOracle(nx, ny, 100.0, semantics=sem, collision=coll).step(steps)
    parts = partition_rows(ny, nslabs)
    slabs = [SlabStandIn(nx, ny, 100.0, r, semantics=sem, collision=coll) for r in parts]
    LocalSlabs(slabs).step(steps)
    fin = np.concatenate([s.fin for s in slabs], axis=2)
    u = np.concatenate([s.u for s in slabs], axis=2)
    rho = np.concatenate([s.rho for s in slabs], axis=1)
    assert np.array_equal(fin, ref.fin) and np.array_equal(u, ref.u) and np.array_equal(rho, ref.rho)


@pytest.mark.parametrize("unit", [3, 5])
@pytest.mark.parametrize("nslabs", [2, 3, 8])
def test_local_slabs_multi_step_units_equal_single_domain(nslabs, unit):
    """The launch-unit schedule with deep halos (S complete rows per side, then S steps without communication) for first,
    middle and last slabs, uneven heights, several step() calls of lengths that leave every kind of remainder."""
    nx, ny = 18, 67
    ref = on.CavityOracle(nx, ny, 100.0, semantics="mrt_gpu", collision="MRT")
    slabs = [SlabStandIn(nx, ny, 100.0, r, semantics="mrt_gpu", collision="MRT", unit=unit) for r in partition_rows(ny, nslabs)]
    drv = LocalSlabs(slabs)
    for n in (1, unit, 2 * unit + 1, 4, unit + 2, 3):
        drv.step(n)
        ref.step(n)
        fin = np.concatenate([s.fin for s in slabs], axis=2)
        u = np.concatenate([s.u for s in slabs], axis=2)
        rho = np.concatenate([s.rho for s in slabs], axis=1)
        assert np.array_equal(fin, ref.fin) and np.array_equal(u, ref.u) and np.array_equal(rho, ref.rho), n


def test_slabs_that_disagree_on_the_unit_are_refused():
    nx, ny = 16, 40
    parts = partition_rows(ny, 2)
    slabs = [SlabStandIn(nx, ny, 100.0, parts[0], semantics="mrt_gpu", collision="MRT", unit=5),
             SlabStandIn(nx, ny, 100.0, parts[1], semantics="mrt_gpu", collision="MRT", unit=3)]
    drv = LocalSlabs(slabs)
    drv.step(1)
    with pytest.raises(RuntimeError, match="disagree"):
        drv.step(10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nx, ny, steps, sem, coll, q, unit=1):
    import torch.distributed as dist
    from latticeboltzmannsimulations_amd.slab import HaloDriver
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        rows = partition_rows(ny, world)[rank]
        st = SlabStandIn(nx, ny, 100.0, rows, semantics=sem, collision=coll, unit=unit)
        drv = HaloDriver(st, rank, world, device="cpu")
        for n in (steps if isinstance(steps, (list, tuple)) else [steps]):
            drv.step(n)
        from latticeboltzmannsimulations_amd.slab import global_mean_u
        q.put((rank, rows, st.fin, st.u, st.rho, global_mean_u(st, world)))     # (one all-reduce of a double: SURVEY 8(e))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,sem,coll,unit,ny,steps", [(2, "mrt_gpu", "MRT", 1, 19, 10), (2, "mrt_py", "SRT", 1, 19, 10), (3, "mrt_gpu", "SRT", 1, 19, 10),
                                                           # multi-step launch units with deep halos, several calls (world 2, 3, 8)
                                                           (2, "mrt_gpu", "MRT", 5, 23, [1, 12, 4]), (3, "mrt_gpu", "MRT", 4, 31, [9, 7]),
                                                           (8, "mrt_gpu", "MRT", 5, 83, [1, 5, 8])])
def test_gloo_halo_driver_equals_single_domain(world, sem, coll, unit, ny, steps):
    import torch.multiprocessing as mp
    nx = 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, ny, steps, sem, coll, q, unit)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    ref = on.CavityOracle(nx, ny, 100.0, semantics=sem, collision=coll).step(sum(steps) if isinstance(steps, list) else steps)
    fin = np.concatenate([g[2] for g in got], axis=2)
    u = np.concatenate([g[3] for g in got], axis=2)
    rho = np.concatenate([g[4] for g in got], axis=1)
    assert np.array_equal(fin, ref.fin) and np.array_equal(u, ref.u) and np.array_equal(rho, ref.rho)
    # the convergence quantity of MRT_GPU.py:883 from the slabs: every rank got the same whole-lattice mean
    means = [g[5] for g in got]
    assert max(means) - min(means) < 1e-15 and abs(means[0] - float(ref.u.mean())) < 1e-12
