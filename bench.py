#!/usr/bin/env python3
"""bench.py -- MLUPS of the D2Q9 MRT lid-driven-cavity step on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one fused pull-stream + collide pass over the whole lattice (all slabs).
Workload (default): BASELINE.json configs[2] / north_star target -- 4096 x 4096 D2Q9 MRT cavity,
Re = 1000, fp32, MRT_GPU.py semantics; for N > 1 that SAME lattice is cut into N y-slabs, one per rank
(strong scaling, north_star: "4096^2 ... at 1/2/4/8 MI355X"), which exchange halos with their neighbours by
RCCL send/recv inside lbm_step().  --config c4 (8192^2 fp64, Re 3200) is cut the same way; --config c5 is the
weak series of configs[4]: every rank holds 16384 x 2048 of a 16384 x (2048 N) lattice, with omega fixed
from the 8-GPU height 16384 (MRT_GPU.py:63 uses the global ysize).  Synthetic input: the reference's own initial
state (equilibrium at rho = 1 with the moving lid, MRT_GPU.py:262-267), built on the device.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (the binding physical limit of the
dominant kernel, frac <= 1, formulas in DESIGN.md section 6) and `cpu_baseline` (the C oracle timed on
this host's cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

SIMDS = 256 * 4          # CUs x SIMDs
CLOCK_PEAK_GHZ = 2.4     # max shader clock (same guide); the chip holds ~2.0-2.2 GHz under these kernels

CONFIGS = {
    # name: (nx, ny_per_gpu, Re, dtype, RT, semantics, label)
    "c3": (4096, 4096, 1000.0, "float32", "MRT", "mrt_gpu", "BASELINE configs[2]: 4096x4096 D2Q9 MRT cavity Re=1000 fp32"),
    "c2": (1024, 1024, 1000.0, "float64", "MRT", "mrt_gpu", "BASELINE configs[1]: 1024x1024 D2Q9 MRT cavity Re=1000 fp64"),
    "c3f64": (4096, 4096, 1000.0, "float64", "MRT", "mrt_gpu", "4096x4096 D2Q9 MRT cavity Re=1000 fp64"),
    "c4": (8192, 8192, 3200.0, "float64", "MRT", "mrt_gpu", "BASELINE configs[3]: 8192x8192 D2Q9 MRT cavity Re=3200 fp64"),
    "c5": (16384, 2048, 5000.0, "float32", "MRT", "mrt_gpu", "BASELINE configs[4]: 16384 wide, 2048 rows per GPU, Re=5000 fp32, omega of the 16384-row lattice"),
}
DEFAULT_SCALING = {"c2": "strong", "c3": "strong", "c3f64": "strong", "c4": "strong", "c5": "weak"}   # as BASELINE.json words them
OMEGA_HEIGHT = {"c5": 16384}    # the weak series keeps the relaxation rates of the 8-GPU lattice


def host_cores():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota (the GPU
    box exposes 256 logical CPUs but grants a 16-CPU share per GPU)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return int(os.environ.get("LBM_BENCH_CPU_THREADS", min(n, 32)))


def cpu_baseline(nx, ny, Re, dtype, RT, semantics, budget_s=15.0, table_s=0.0):
    """The C oracle (oracle/lbm_ref.c, kind 'port') on this host's cores, bounded sample."""
    from oracle import lbm_ref
    threads = max(1, min(lbm_ref.max_threads(), host_cores()))
    lbm_ref.set_threads(threads)
    o = lbm_ref.CavityOracleC(nx, ny, Re, semantics=semantics, collision=RT, dtype=np.dtype(dtype))
    o.step(2)                                   # first touch, thread pool warm
    t = time.perf_counter(); o.step(2); one = (time.perf_counter() - t) / 2
    n = int(max(2, min(200, budget_s / max(one, 1e-6))))
    t = time.perf_counter(); o.step(n); dt = time.perf_counter() - t
    lbm_ref.set_threads(1)
    out = {"value": nx * ny * n / dt / 1e6, "unit": "MLUPS", "cores": threads, "kind": "port",
           "sample": f"{n} steps of the same {nx}x{ny} {dtype} {RT} lattice with the C oracle (oracle/lbm_ref.c, "
                     f"OpenMP, {threads} threads), {dt:.1f} s"}
    if table_s > 0:
        out["table"] = cpu_baseline_table(nx, ny, Re, dtype, RT, semantics, table_s)
    return out


def cpu_baseline_table(nx, ny, Re, dtype, RT, semantics, budget_s=3.0):
    """SURVEY 8(d)'s other CPU figures beside `value`, each bounded to about `budget_s` seconds: the C restatement on the same
    lattice at 1 thread and at 4 (the count hard-coded in the reference's Cython path, functions.pyx:69), and the NumPy restatement
    of MRT.py (its own semantics: SRT, fp64, 1 thread) at 1024^2 (MRT.py's pure-NumPy loop, MRT.py:286-453)."""
    from oracle import lbm_ref
    from oracle.lbm_numpy import CavityOracle
    rows = []
    for threads in (1, 4):
        if threads > max(1, lbm_ref.max_threads()):
            continue
        lbm_ref.set_threads(threads)
        o = lbm_ref.CavityOracleC(nx, ny, Re, semantics=semantics, collision=RT, dtype=np.dtype(dtype))
        t = time.perf_counter(); o.step(1); one = time.perf_counter() - t      # (includes the first touch)
        n = int(max(1, min(50, (budget_s - one) / max(one, 1e-6))))
        t = time.perf_counter(); o.step(n); dt = time.perf_counter() - t
        rows.append({"impl": "oracle/lbm_ref.c", "threads": threads, "lattice": [nx, ny], "case": f"{semantics} {RT} {dtype}",
                     "MLUPS": round(nx * ny * n / dt / 1e6, 2), "steps": n})
    lbm_ref.set_threads(1)
    m = 1024
    o = CavityOracle(m, m, 1000.0, semantics="mrt_py", collision="SRT", dtype=np.float64)
    t = time.perf_counter(); o.step(1); one = time.perf_counter() - t
    n = int(max(1, min(50, (budget_s - one) / max(one, 1e-6))))
    t = time.perf_counter(); o.step(n); dt = time.perf_counter() - t
    rows.append({"impl": "oracle/lbm_numpy.py", "threads": 1, "lattice": [m, m], "case": "mrt_py SRT float64 (what MRT.py runs)",
                 "MLUPS": round(m * m * n / dt / 1e6, 2), "steps": n})
    return rows


def cpu_table():
    """SURVEY 8(d) CPU baselines beside the GPU number: the NumPy restatement (1 thread) and the C restatement at 1, 4 (the
    thread count hard-coded in functions.pyx:69) and all granted cores, MRT.py semantics / SRT / fp64 (what MRT.py runs) and
    MRT_GPU.py semantics / MRT.  Part of the cpu_baseline leg: oracle code timed as a baseline, nothing shipped."""
    from oracle import lbm_ref
    from oracle.lbm_numpy import CavityOracle
    rows = []
    for n, steps in ((128, 200), (1024, 10)):
        o = CavityOracle(n, n, 1000.0, semantics="mrt_py", collision="SRT", dtype=np.float64)
        o.step(2)
        t = time.perf_counter(); o.step(steps); dt = time.perf_counter() - t
        rows.append({"impl": "oracle/lbm_numpy.py", "threads": 1, "lattice": n, "case": "mrt_py SRT f64", "MLUPS": round(n * n * steps / dt / 1e6, 2)})
        print(json.dumps(rows[-1]), flush=True)
    allc = max(1, min(lbm_ref.max_threads(), host_cores()))
    for threads in sorted({1, 4, allc}):
        lbm_ref.set_threads(threads)
        for n, steps in ((128, 2000), (1024, 60), (4096, 6)):
            for sem, coll, dt_ in (("mrt_py", "SRT", np.float64), ("mrt_gpu", "MRT", np.float64), ("mrt_gpu", "MRT", np.float32)):
                o = lbm_ref.CavityOracleC(n, n, 1000.0, semantics=sem, collision=coll, dtype=np.dtype(dt_))
                o.step(2)
                t = time.perf_counter(); o.step(steps); dt = time.perf_counter() - t
                rows.append({"impl": "oracle/lbm_ref.c", "threads": threads, "lattice": n, "case": f"{sem} {coll} {np.dtype(dt_).name}",
                             "MLUPS": round(n * n * steps / dt / 1e6, 2)})
                print(json.dumps(rows[-1]), flush=True)
    lbm_ref.set_threads(1)
    return rows


def slab_of(config, scaling, world, rank):
    """(global height, (first row, rows)) of rank `rank`: weak scaling gives every rank the N = 1 workload as a y-slab of a
    lattice `world` times as tall; strong scaling cuts the N = 1 lattice into `world` slabs."""
    from latticeboltzmannsimulations_amd.slab import partition_rows
    ny_gpu = CONFIGS[config][1]
    if scaling == "weak":
        return ny_gpu * world, (rank * ny_gpu, ny_gpu)
    return ny_gpu, partition_rows(ny_gpu, world)[rank]


def make_solver(config, scaling, world, rank, dev, kernel, arith, tuning=None):
    """The CavitySolver of rank `rank` for a BASELINE configuration (whole lattice when world == 1)."""
    from latticeboltzmannsimulations_amd import CavitySolver, relaxation
    from latticeboltzmannsimulations_amd.slab import partition_rows
    nx, ny_gpu, Re, dtype, RT, sem, _ = CONFIGS[config]
    NY, rows = slab_of(config, scaling, world, rank)
    mr = min(n for _, n in partition_rows(NY, world)) if world > 1 else None
    s = CavitySolver(nx, NY, Re, RT=RT, semantics=sem, dtype=np.dtype(dtype), device=dev, rows=rows if world > 1 else None,
                     kernel=kernel, arith=arith, min_rows=mr, tuning=tuning)
    if config in OMEGA_HEIGHT:      # relaxation rates of the full-size lattice of the series, whatever this run's height
        s.relax = relaxation(Re, OMEGA_HEIGHT[config], s.uLB, 1.2, 1.2)
        s.set_relaxation(0, **s.relax)
    return s, NY, rows


PROTOCOL_KEYS = ("kernel", "steps_per_launch", "frame", "deep_halo", "stream", "vec", "layout", "units")


def rank_plan(config, scaling, world, rank, kernel, arith, steps, tuning=None):
    """Dry run (lbm_plan: host logic of the library, no GPU) of the launch plan rank `rank` would derive for this bench configuration:
    the dict of CavitySolver.describe() + `units`, the launch units of `steps` steps from a fresh lattice.  The arguments go through
    the same slab_of / partition_rows / min_rows path as make_solver."""
    from latticeboltzmannsimulations_amd import launch_plan
    from latticeboltzmannsimulations_amd.slab import partition_rows
    nx, ny_gpu, Re, dtype, RT, sem, _ = CONFIGS[config]
    NY, rows = slab_of(config, scaling, world, rank)
    mr = min(n for _, n in partition_rows(NY, world)) if world > 1 else None
    return launch_plan(nx, NY, Re, steps=steps, RT=RT, semantics=sem, dtype=np.dtype(dtype), rows=rows if world > 1 else None,
                       kernel=kernel, arith=arith, min_rows=mr, tuning=tuning)


def check_plans(config, scaling, world, kernel, arith, steps, tuning=None):
    """Pre-flight of a multi-rank run: the plans of ALL ranks (each rank can compute every rank's: pure host logic) must agree on
    everything that shapes the exchange protocol -- neighbours post matching send / receive sequences.  Returns (plans, None) or
    (plans, "rank r: key a != b")."""
    plans = [rank_plan(config, scaling, world, r, kernel, arith, steps, tuning) for r in range(world)]
    for r in range(1, world):
        for k in PROTOCOL_KEYS:
            if plans[r].get(k) != plans[0].get(k):
                return plans, f"rank {r}: {k} = {plans[r].get(k)!r}, rank 0: {plans[0].get(k)!r}"
    return plans, None


def unit_plan(solver, steps):
    """The launch units lbm_step(steps) will run from the solver's current (not just uploaded) state: [S1, S2, ...]."""
    out, left = [], steps
    while left > 0:
        S = solver.next_unit(left)
        out.append(S)
        left -= S
    return out


def timed_rate(config, dev, kernel, arith, steps, warm):
    """MLUPS and ms per step of a whole BASELINE configuration on one GPU, HIP events around `steps` steps (best of two)."""
    s, NY, _ = make_solver(config, "strong", 1, 0, dev, kernel, arith)
    try:
        s.step(warm); s.sync()
        ms = min(s.time_steps(steps) for _ in range(2)) / steps
        return {"MLUPS": round(CONFIGS[config][0] * NY / ms / 1e3, 1), "ms_per_step": round(ms, 5), "steps": steps,
                "steps_per_launch": s.next_unit(1000)}
    finally:
        s.close()


def load_static(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-table", action="store_true", help="time the CPU restatements only (SURVEY 8d table) and exit")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--arith", choices=["fast", "strict"], default="fast",
                    help="fast: MRT operator in factored form with fused multiply-adds (agrees with the oracle to rounding, "
                         "tests/test_gpu_parity.py::test_fast_arithmetic_*); strict: the reference's operation order "
                         "(bit-identical to the oracle), also measured and reported under other.strict_arith")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto",
                    help="auto: as BASELINE.json words the configuration (c3 / c4: strong -- one lattice cut into N slabs; c5: weak)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short side measurements reported under 'other'")
    a = ap.parse_args()
    if a.cpu_table:
        cpu_table()
        return
    scaling = DEFAULT_SCALING[a.config] if a.scaling == "auto" else a.scaling

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 "
                     "--master-port 29500 bench.py --gpus %d ..." % (a.gpus, a.gpus))
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    # (multi-process GPU work on this platform: the host driver only supports dmabuf IPC -- without this RCCL's handle exchange fails with
    # "hipIpcGetMemHandle: invalid argument"; the launcher's environment normally carries it already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch                      # plumbing: process group, barrier, device sync
    import torch.distributed as dist
    from latticeboltzmannsimulations_amd.slab import attach_rccl

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no HIP device visible)")
    ndev = torch.cuda.device_count()
    dev = local_rank if local_rank < ndev else 0       # a launcher may expose one GPU per rank
    if world > 1 and ndev < world and ndev != 1:
        sys.exit(f"{world} ranks but {ndev} visible GPUs")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))

    nx, ny_gpu, Re, dtype, RT, sem, label = CONFIGS[a.config]
    if world > 1:
        # Pre-flight, before any rank creates its lattice: every rank derives the launch plan of EVERY rank (host logic only) and
        # stops the job, naming the differing item, if two ranks would run different protocols.
        plans, bad = check_plans(a.config, scaling, world, a.kernel, a.arith, a.warmup + a.steps)
        if bad:
            print(f"[rank {rank}] launch plans differ between ranks -- {bad}", file=sys.stderr, flush=True)
            sys.exit(3)
    solver, NY, rows = make_solver(a.config, scaling, world, rank, dev, a.kernel, a.arith)
    driver = None                                 # (set only by the fallback below)
    if world > 1:
        print(f"[rank {rank}] device {dev} rows {rows} plan: {solver.describe()}", file=sys.stderr, flush=True)
        ok = 1
        try:
            attach_rccl(solver, rank, world)      # lbm_comm_init: communicator + the plan handshake with both neighbours
        except RuntimeError as e:                 # LBM_ERR_STATE names the item a neighbour plans differently
            print(f"[rank {rank}] lbm_comm_init failed: {e}", file=sys.stderr, flush=True)
            if "launch plan" in str(e):
                sys.exit(4)                       # (a protocol mismatch: no transport would fix it)
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t[0]) == 1:
            print(f"[rank {rank}] communicator up, plan handshake with the neighbours ok", file=sys.stderr, flush=True)
        else:
            # The in-library communicator could not be created on some rank (it has never met a second GPU: DESIGN 7).  Rather than no
            # number at all: the same launch units with the halos moved by torch.distributed's own RCCL communicator between the units
            # (slab.HaloDriver, the externally driven path the GPU tests run on one device) -- NOT overlapped with the bulk launch, and
            # said so in config.halo.
            from latticeboltzmannsimulations_amd.slab import HaloDriver
            print(f"[rank {rank}] falling back to halos driven through torch.distributed between the launch units", file=sys.stderr, flush=True)
            try:
                solver.close()
                solver, NY, rows = make_solver(a.config, scaling, world, rank, dev, a.kernel, a.arith)
                driver = HaloDriver(solver, rank, world, device="cuda")
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] fallback failed too: {e}", file=sys.stderr, flush=True)
                sys.exit(4)

    if world == 1 and os.environ.get("LBM_BENCH_FORCE_FALLBACK"):     # (test hook: the fallback's stepping path on one GPU)
        from latticeboltzmannsimulations_amd.slab import HaloDriver
        driver = HaloDriver(solver, 0, 1, device="cuda")

    def fence():
        solver.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Device wake-up, identical for every N: ~40 ms of plain HBM copies so that clocks and power state are those of a
    # loaded device before anything is timed (a cold first 200-step window reads ~10 % low: profiles/r01_logs/perf12.log).
    # These are not workload steps; the W warm-up steps of the contract follow.
    wake_gbps = solver.copy_bandwidth(1 << 30, 100)
    # ... and ~20 ms of packed fp32 FMAs (lbm_fma_rate): the copies leave the core clock where a memory-bound load needs it, the
    # streaming kernel is as much arithmetic- as memory-bound, and its first launches after copies alone run 6 % slower than in a
    # long run (tools/probes/wakeup_ab.py, profiles/r02_logs/wakeup_ab.log).  Not workload steps either.
    wake_tflops = solver.fma_rate(20.0)
    (driver.step if driver else solver.step)(a.warmup)
    fence()
    units = unit_plan(solver, a.steps)        # the launches the K timed steps consist of
    t0 = time.perf_counter()
    if driver:
        driver.step(a.steps)
        ev_ms = float("nan")
    else:
        ev_ms = solver.time_steps(a.steps)  # HIP events on the compute stream around the K steps
    fence()
    dt = time.perf_counter() - t0
    if driver:
        ev_ms = dt * 1e3
    if world > 1:
        t = torch.tensor([dt, ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])

    cells_total = nx * NY
    cells_rank = nx * rows[1]
    es = np.dtype(dtype).itemsize
    mlups = cells_total * a.steps / dt / 1e6
    step_ms = ev_ms / a.steps
    # algorithmic bytes per lattice update = read 9 + write 9 populations = 18 * sizeof(real).
    alg_bytes_step = cells_rank * 2 * 9 * es
    algorithmic_GBps = alg_bytes_step / (step_ms * 1e-3) / 1e9

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: a launch unit of S steps = ONE launch of k_stepS_deep (tiles + wall frame) ------
        # Its own minimum HBM traffic is one read and one write of the lattice per LAUNCH (2 * 9 * sizeof(real) * cells), whatever
        # S: that is the byte count of the HBM line below, so frac <= 1 by construction; `algorithmic_GBps` keeps SURVEY 8(d)'s
        # per-update figure (which a temporally blocked kernel can exceed).  The launch duration is the HIP-event time of the
        # timed region divided by its launch units (with the default plan every unit is the same kernel at the same S; single
        # steps, if the step count leaves a remainder, are counted as units of their own).
        S_dom = max(units) if units else 1
        pure = bool(units) and all(u == S_dom for u in units)
        launch_ms = ev_ms / len(units) if units else float("nan")
        region_launch_ms = launch_ms
        if units and not pure and world == 1:
            # The timed region mixes launch units (e.g. 8 + 8 + 4 steps, the last one on another kernel): its average is not the
            # duration of the dominant kernel's launch.  Measure that one on its own, right after the timed region: three launches
            # of S_dom steps, HIP events around them (outside the timed steps, so `value` is untouched).
            solver.step(2 * S_dom)          # (the device idled while the host got here: two launches bring it back to speed)
            launch_ms = min(solver.time_steps(3 * S_dom) for _ in range(3)) / 3
        min_bytes_launch = cells_rank * 2 * 9 * es
        hbm_achieved = min_bytes_launch / (launch_ms * 1e-3) / 1e9
        plan = solver.describe()
        kname = plan["kernel"] if a.kernel in ("auto", "tb", "stream") and plan["kernel"] != "none" else \
            {"vec": "k_step_vec", "generic": "k_step_generic", "push": "k_push_collide+k_push_bc"}.get(a.kernel, a.kernel)
        key = f"{a.config}:{world}:{a.kernel}:{a.arith}"
        tr = load_static("traffic.json").get(key, {})
        traffic = None
        if tr and tr.get("steps_per_launch", S_dom) == S_dom and tr.get("kernel", kname) == kname:   # PMC bytes of THIS kernel at THIS S, from the
            traffic = tr.get("hbm_bytes_per_launch")                                                 # committed profile (else null)
        vm = load_static("valu_mix.json").get(f"{kname}:{dtype}:{RT}:{a.arith}:turb0", {})
        valu = None
        if vm and plan.get("wave_updates", 0) > 0 and S_dom == plan["steps_per_launch"]:
            # VALU issue line: (wave, level) updates of one launch (lbm_describe: rows x strips x levels, lead rows included) x the
            # issue cycles of one update (the kernel's level loop priced per instruction class: tools/valu_mix.py with the costs
            # measured by tools/valu_issue.hip), over the cycles the 1024 SIMDs offer during the launch at the 2.4 GHz peak clock
            cyc = vm["issue_cycles_per_wave_update"] * plan["wave_updates"]
            valu = {"wave_updates_per_launch": plan["wave_updates"], "issue_cycles_per_wave_update": vm["issue_cycles_per_wave_update"],
                    "frac": round(cyc / (launch_ms * 1e-3 * CLOCK_PEAK_GHZ * 1e9 * SIMDS), 4), "peak": f"{SIMDS} SIMDs x {CLOCK_PEAK_GHZ} GHz",
                    "source": vm.get("source")}
        roof = {"bound": "hbm", "achieved": round(hbm_achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(hbm_achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "kernel": kname, "steps_per_launch": S_dom, "launches_timed": len(units), "launch_ms": round(launch_ms, 5),
                "launch_ms_source": "HIP events over the timed region / its launches" if pure or world > 1 else
                                    "three launches of the dominant kernel timed right after the timed region (the region mixes unit lengths; "
                                    f"its own average per launch unit: {region_launch_ms:.5f} ms)",
                "bytes_per_launch": min_bytes_launch,
                "traffic_static": True if traffic is not None else None,
                "traffic_source": tr.get("source") if traffic is not None else None,
                "traffic_GBps": None if traffic is None else round(traffic / (launch_ms * 1e-3) / 1e9, 1),
                "traffic_frac_of_peak": None if traffic is None else round(traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "valu_issue": valu,
                "algorithmic_GBps": round(algorithmic_GBps, 1), "algorithmic_bytes_per_step": alg_bytes_step,
                "event_ms_per_step": round(step_ms, 5), "units": units if len(units) <= 12 else units[:6] + ["..."] + units[-3:],
                "note": "achieved = (one read + one write of the lattice = the kernel's minimum HBM bytes per launch) / launch duration "
                        "(HIP events over the timed region / launches); one launch advances steps_per_launch time steps through LDS, so "
                        "SURVEY 8(d)'s per-update figure is reported separately as algorithmic_GBps = 72 or 144 B x updates / time and is "
                        "not a bound; traffic = PMC FETCH_SIZE x 2 + WRITE_SIZE of the same kernel from the committed rocprofv3 profile "
                        "(static, not measured in this run)" + ("" if pure else "; the timed region mixes launch units of different length")}
        # the nearer of the two limits (informational; `bound` / `frac` stay the HBM line the contract asks for)
        roof["binding"] = "valu_issue" if valu and valu["frac"] > max(roof["frac"], roof["traffic_frac_of_peak"] or 0) else "hbm"
        other = {}
        if not a.no_extra and world == 1:
            try:
                other["device_copy_GBps"] = round(wake_gbps, 1)
                other["device_fma_TFLOPs"] = round(wake_tflops, 1)
                # the same workload advanced ONE step per launch (k_step_vec): the HBM-streaming reference point
                s1, _, _ = make_solver(a.config, scaling, 1, 0, dev, "vec", "strict")
                with s1:
                    s1.step(10); s1.sync()
                    ms1 = s1.time_steps(50) / 50
                other["one_step_per_launch"] = {"MLUPS": round(cells_total / ms1 / 1e3, 1), "ms_per_step": round(ms1, 5),
                                                "hbm_GBps": round(alg_bytes_step / ms1 / 1e6, 1),
                                                "frac_of_peak": round(alg_bytes_step / ms1 / 1e6 / HBM_PEAK_GBPS, 4)}
                # the reference's own structure on this device: collide-and-push + wall-rule / copy kernel, two launches per step
                s0, _, _ = make_solver(a.config, scaling, 1, 0, dev, "push", "strict")
                with s0:
                    s0.step(10); s0.sync()
                    ms0 = s0.time_steps(40) / 40
                other["reference_scheme_push"] = {"MLUPS": round(cells_total / ms0 / 1e3, 1), "ms_per_step": round(ms0, 5)}
                if a.arith == "fast":     # the same workload in the reference's exact operation order (bit-identical to the oracle),
                    s2, _, _ = make_solver(a.config, scaling, 1, 0, dev, a.kernel, "strict")    # same step count as the headline
                    with s2:
                        s2.step(max(a.warmup, 40)); s2.sync()
                        ms2 = min(s2.time_steps(a.steps) for _ in range(2)) / a.steps
                        other["strict_arith"] = {"MLUPS": round(cells_total / ms2 / 1e3, 1), "ms_per_step": round(ms2, 5),
                                                 "steps": a.steps, "steps_per_launch": s2.next_unit(1000),
                                                 "note": "reference operation order, bit-identical to the oracle"}
                if a.steps < 400:   # the same workload over a long window (the driver's K is a 1 ms burst): 800 steps, HIP events
                    msl = solver.time_steps(800) / 800
                    other["long_run"] = {"MLUPS": round(cells_total / msl / 1e3, 1), "ms_per_step": round(msl, 5), "steps": 800}
                # the other BASELINE configurations on this one GPU (whole lattice; c5: the per-GPU slab of the weak series)
                for cfg, n in (("c2", 200), ("c3f64", 100), ("c4", 50), ("c5", 100)):
                    if cfg == a.config:
                        continue
                    other[cfg] = {"workload": CONFIGS[cfg][6]}
                    for ar in ("fast", "strict"):
                        other[cfg][ar] = timed_rate(cfg, dev, "auto", ar, n, 20)
                # the headline input is the prescribed rest state; the same measurement on populations with +-1e-3
                # relative noise (numpy default_rng(0), fp32 noise field), same wake-up, warm-up and step count: the rate does
                # not depend on the data (profiles/r01_logs/clock_probe.log: 61.2 vs 61.4 us per step over 30 000 steps, the
                # package at its 1.4 kW cap either way; earlier, lower "noisy" figures were cold-clock artefacts)
                _, _, fin = solver.get_fields(want_fin=True)
                rng = np.random.default_rng(0)
                for k in range(9):
                    fin[k] *= (1.0 + 1e-3 * rng.standard_normal(fin[k].shape, dtype=np.float32)).astype(fin.dtype)
                solver.set_state(fin)
                del fin
                solver.copy_bandwidth(1 << 30, 100)     # the device idled while the host built the noise: wake it up again,
                solver.fma_rate(20.0)                   # the same way as before the headline measurement
                solver.step(max(a.warmup, 6)); solver.sync()
                msn = solver.time_steps(a.steps) / a.steps
                other["noisy_state"] = {"MLUPS": round(cells_total / msn / 1e3, 1), "ms_per_step": round(msn, 5)}
            except Exception as e:      # measurement nicety only
                other["error"] = str(e)
        out = {
            "metric": "MLUPS (million lattice updates/sec) D2Q9 MRT cavity",
            "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 5), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32" if dtype == "float32" else "f64", "data": "synthetic",
            "config": {"workload": label + (f"; {world} y-slabs of {nx}x{rows[1]}, lattice {nx}x{NY}" if world > 1 else ""),
                       "lattice": [nx, NY], "Re": Re, "collision": RT, "semantics": sem, "kernel": a.kernel,
                       "arith": a.arith + (" (factored MRT operator + fused multiply-adds; same operator, agrees with the strict path "
                                           "to rounding, not bit for bit; the reference-order path is other.strict_arith)" if a.arith == "fast"
                                           else " (reference operation order, bit-identical to the oracle)"),
                       "halo": ("none" if world == 1 else
                                f"FALLBACK (lbm_comm_init failed): torch.distributed send/recv of {S_dom} complete rows per side between the "
                                f"{S_dom}-step launch units, not overlapped" if driver else
                                f"rccl send/recv inside lbm_step: {S_dom} complete rows per side before each {S_dom}-step launch, overlapped"),
                       "device_wakeup": "before the W warm-up steps, identical for every N, not workload steps: 100 device copies of 1 GiB "
                                        "(~40 ms) + ~20 ms of packed fp32 FMAs (lbm_fma_rate), so that memory AND core clocks are those of "
                                        "a loaded device when the warm-up starts"},
            "roofline": roof,
        }
        if other:
            out["other"] = other
    solver.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nx, NY, Re, dtype, RT, sem, table_s=3.0)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
