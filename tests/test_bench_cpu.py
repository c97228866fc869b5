"""CPU checks of bench.py's host-side pieces (no GPU): configs table, CPU-share detection, the cpu_baseline leg."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_configs_follow_baseline_json():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "4096" in base["configs"][2] and "fp32" in base["configs"][2]
    nx, ny, Re, dtype, RT, sem, _ = bench.CONFIGS["c3"]
    assert (nx, ny, Re, dtype, RT, sem) == (4096, 4096, 1000.0, "float32", "MRT", "mrt_gpu")
    assert bench.CONFIGS["c2"][:4] == (1024, 1024, 1000.0, "float64")
    assert bench.CONFIGS["c5"][:2] == (16384, 2048)      # 16384^2 / 8 GPUs


def test_host_cores_is_positive_and_bounded():
    n = bench.host_cores()
    assert 1 <= n <= 32


def test_cpu_baseline_leg_runs_on_a_small_sample():
    r = bench.cpu_baseline(128, 128, 1000.0, "float32", "MRT", "mrt_gpu", budget_s=0.3)
    assert r["kind"] == "port" and r["unit"] == "MLUPS" and r["value"] > 0.1 and r["cores"] >= 1


def test_wrong_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env)
    assert p.returncode != 0 and "torch.distributed.run" in (p.stderr + p.stdout)


def test_traffic_table_has_the_default_bench_entry():
    """bench.py fills roofline.traffic from profiles/traffic.json (PMC-measured HBM bytes per launch of the dominant kernel, keyed
    by configuration and tagged with the steps per launch they were measured at; bench.py drops an entry whose tag differs)."""
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key in ("c3:1:auto:fast", "c3:1:auto:strict", "c3:1:vec:strict", "c3:1:tb:fast"):
        assert t[key]["hbm_bytes_per_launch"] > 0 and t[key]["steps_per_launch"] >= 1 and t[key]["source"] and t[key]["kernel"], key
    assert t["c3:1:auto:fast"]["kernel"] == "k_stream_walls" and t["c3:1:auto:fast"]["steps_per_launch"] == 8
    # a multi-step launch moves about one read + one write of the lattice; so does a single step
    for key in ("c3:1:auto:fast", "c3:1:auto:strict", "c3:1:vec:strict", "c3:1:tb:fast"):
        assert 0.95 < t[key]["hbm_bytes_per_launch"] / (4096 * 4096 * 72) < 1.5, key


def test_slab_plan_of_the_multi_gpu_bench():
    """bench.py --gpus N: strong scaling (default for c3 / c4) = the N = 1 lattice cut into N slabs; weak scaling (c5) = N slabs
    of the N = 1 workload stacked in y, relaxation rates of the 8-GPU lattice."""
    assert bench.DEFAULT_SCALING["c3"] == "strong" and bench.DEFAULT_SCALING["c4"] == "strong" and bench.DEFAULT_SCALING["c5"] == "weak"
    for world in (1, 2, 4, 8):
        rows = [bench.slab_of("c3", "weak", world, r) for r in range(world)]
        assert all(NY == 4096 * world for NY, _ in rows)
        assert [r[1] for r in rows] == [(4096 * i, 4096) for i in range(world)]
        for cfg, ny in (("c3", 4096), ("c4", 8192)):
            rows = [bench.slab_of(cfg, "strong", world, r) for r in range(world)]
            assert all(NY == ny for NY, _ in rows)
            assert rows[0][1][0] == 0 and sum(r[1][1] for r in rows) == ny
            assert all(rows[i][1][0] + rows[i][1][1] == rows[i + 1][1][0] for i in range(world - 1))
    NY, (y0, n) = bench.slab_of("c5", "weak", 8, 7)
    assert (NY, y0, n) == (16384, 14336, 2048)                   # BASELINE configs[4]: 16384 x 16384 over 8 GPUs
    assert bench.OMEGA_HEIGHT["c5"] == 16384


def test_cpu_baseline_table_is_bounded_and_complete():
    """cpu_baseline.table of the driver line: the C restatement at 1 and 4 threads (functions.pyx:69) on the same
    lattice and the NumPy restatement of MRT.py at 1 thread, each bounded."""
    import time
    t = time.time()
    rows = bench.cpu_baseline_table(256, 256, 1000.0, "float32", "MRT", "mrt_gpu", budget_s=0.2)
    assert time.time() - t < 30
    assert [r["threads"] for r in rows if r["impl"] == "oracle/lbm_ref.c"] in ([1, 4], [1])
    assert rows[-1]["impl"] == "oracle/lbm_numpy.py" and rows[-1]["threads"] == 1 and all(r["MLUPS"] > 0 for r in rows)


def test_every_rank_of_the_multi_gpu_bench_plans_the_same_protocol():
    """bench.py --gpus N for N = 2, 4, 8 x {c3, c4, c5}, driven through bench.py's own slab_of / partition_rows /
    min_rows argument path, with the LIBRARY's plan logic (lbm_plan: lbm_create's launch plan + lbm_next_unit's units, no device):
    every rank must derive the same kernel, steps per launch, frame width, deep halo and the same sequence of launch units -- they post
    matching send / receive sequences (lbm_comm_init cross-checks the same items at run time).  Also: first / middle / last rank get
    the slab flag, N = 1 does not; an uneven cut (neighbours on opposite sides of a size threshold) agrees through
    min_rows, and WITHOUT min_rows the check catches the disagreement."""
    from latticeboltzmannsimulations_amd import launch_plan
    from latticeboltzmannsimulations_amd.slab import partition_rows
    for cfg in ("c3", "c4", "c5"):
        scaling = bench.DEFAULT_SCALING[cfg]
        for world in (2, 4, 8):
            for arith in ("fast", "strict"):
                plans, bad = bench.check_plans(cfg, scaling, world, "auto", arith, 25)
                assert bad is None, (cfg, world, arith, bad)
                assert [q["slab"] for q in plans] == [1] * world and sum(plans[0]["units"]) == 25 and plans[0]["units"][0] == 1
                assert plans[0]["deep_halo"] == 1 and plans[0]["steps_per_launch"] >= 3, (cfg, world, plans[0])
                # the figures the protocol uses are those of the smallest slab
                NY = bench.slab_of(cfg, scaling, world, 0)[0]
                assert min(n for _, n in partition_rows(NY, world)) >= 2 * plans[0]["frame"]
    one = bench.rank_plan("c3", "strong", 1, 0, "auto", "fast", 20)
    assert one["slab"] == 0 and one["kernel"] == "k_stream_walls" and one["units"] == [1, 8, 8, 3]
    # 768 x 1535 rows cut in two: 768 and 767 rows sit on opposite sides of the tile kernel's size threshold
    parts = partition_rows(1535, 2)
    with_min = [launch_plan(768, 1535, 1000.0, steps=12, rows=r, min_rows=min(n for _, n in parts)) for r in parts]
    assert all(with_min[0][k] == with_min[1][k] for k in bench.PROTOCOL_KEYS)
    without = [launch_plan(768, 1535, 1000.0, steps=12, rows=r) for r in parts]
    assert any(without[0][k] != without[1][k] for k in bench.PROTOCOL_KEYS), "the example no longer straddles a threshold: pick another"


def test_a_short_slab_leaves_cus_to_its_band_workgroups_per_shader_engine():
    """A slab's unit is an edge launch (the interface bands: 2 x nstrips workgroups, each holding a CU) and a bulk launch.  Workgroups are
    dealt to the eight XCDs and their four shader engines in turn, so a one-round bulk launch must leave the band workgroups room in
    every shader engine -- ceil(bands / 32) + ceil(bulk / 32) <= 8 -- up to a bulk launch of three band workgroups' length (lbm_plan.hip,
    plan_stream; profiles/r03_logs/slab_walls.log); a long or multi-round bulk launch takes the whole device."""
    from latticeboltzmannsimulations_amd import launch_plan
    for rows, reserve in ((512, True), (1024, True), (2048, False)):
        d = launch_plan(4096, 3 * rows, 1000.0, dtype=np.float32, arith="fast", rows=(rows, rows), steps=16)
        assert d["kernel"] == "k_stream_walls" and d["slab"] == 1 and d["frame"] == 8, d
        bands = 2 * 17                                                # two interfaces x 17 strips of 240 columns
        fits = -(-bands // 32) + -(-d["workgroups"] // 32) <= 8
        assert fits == reserve, (rows, d["workgroups"])
    first = launch_plan(4096, 3072, 1000.0, dtype=np.float32, arith="fast", rows=(0, 1024), steps=16)        # one interface: 17 band workgroups
    assert -(-17 // 32) + -(-first["workgroups"] // 32) <= 8 and first["workgroups"] > 187, first
    wide = launch_plan(16384, 3 * 2048, 1000.0, dtype=np.float32, arith="fast", rows=(2048, 2048), steps=16)  # several rounds: no reserve
    assert wide["workgroups"] > 256, wide
