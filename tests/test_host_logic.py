"""CPU tests of the host-side logic around the hot path (rows a2, a10, a11 of SURVEY.md 8):
relaxation parameters, Ghia table + the reference's metrics, the .vtr writer (byte fixtures
produced by the reference's own VTKWrapper/pyevtk), slab partitioning."""
import json
import os

import numpy as np
import pytest

from latticeboltzmannsimulations_amd import ghia, relaxation
from latticeboltzmannsimulations_amd.VTKWrapper import saveToVTK
from latticeboltzmannsimulations_amd.slab import partition_rows, neighbours

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_relaxation_matches_reference_formulas():
    rec = json.load(open(os.path.join(GOLDEN, "survey_appendix_c.json")))
    r = relaxation(100.0, 128)
    assert r["omega"] == rec["omega"]
    assert r["omega_e"] == 1.0 and r["omega_eps"] == 1.2 and r["omega_q"] == 1.2    # MRT_GPU.py:89-90
    assert relaxation(5000.0, 16384)["omega"] == rec["omega_by_config"]["C5_16384_Re5000"]
    assert relaxation(3200.0, 8192)["omega"] == rec["omega_by_config"]["C4_8192_Re3200"]


def test_ghia_table_matches_reference_csv_fixture():
    g = np.load(os.path.join(GOLDEN, "ghia.npz"))
    assert np.array_equal(ghia.Y_GHIA, g["Y"]) and np.array_equal(ghia.X_GHIA, g["X"])
    assert np.array_equal(ghia.UX_GHIA, g["Ux"]) and np.array_equal(ghia.UY_GHIA, g["Uy"])
    assert np.array_equal(ghia.VORTEX_GHIA, g["vortices"])
    Y, Ux, X, Uy = ghia.ghia_profiles(1000)
    assert Ux[1] == 0.65928 and Uy[1] == -0.21388
    with pytest.raises(KeyError):
        ghia.ghia_profiles(123)
    xv, yv = ghia.ghia_vortices(100)
    assert xv.tolist() == [0.6172, 0.0313, 0.9453] and yv.tolist() == [0.7344, 0.0391, 0.0625]


def test_sample_rows_match_record():
    rec = json.load(open(os.path.join(GOLDEN, "survey_appendix_c.json")))
    assert ghia.sample_rows(128).tolist() == rec["LBMy"]


def test_metrics_on_a_synthetic_field():
    # a field whose middle column reproduces Ghia exactly at the sampled rows -> both metrics ~ 1
    X = Y = 128
    u = np.zeros((2, X, Y))
    _, Ux, _, _ = ghia.ghia_profiles(100)
    rows = ghia.sample_rows(Y)
    u[0, X // 2, rows] = Ux[:-1][::-1] * 0.08
    assert ghia.regression_value(u, 100, 0.08) == pytest.approx(1.0, abs=1e-12)
    assert ghia.r2_value(u, 100, 0.08) > 0.9999
    cx, cy = ghia.centrelines(u, 0.08)
    assert cx.shape == (Y,) and cy.shape == (X,)


def test_regression_value_matches_record_on_oracle_field():
    """MRT.py:559-561 metric on the C1 oracle field equals the value recorded in SURVEY App. C."""
    from oracle.lbm_numpy import CavityOracle
    rec = json.load(open(os.path.join(GOLDEN, "survey_appendix_c.json")))
    o = CavityOracle(128, 128, 100.0, semantics="mrt_py", collision="SRT").step(1000)
    assert ghia.regression_value(o.u, 100, 0.08) == pytest.approx(rec["regression_value"], rel=1e-12)


@pytest.mark.parametrize("fixture", ["vtr_4x3.npz", "vtr_5x4_f32.npz"])
def test_vtr_bytes_equal_reference_writer(fixture, tmp_path, monkeypatch):
    g = np.load(os.path.join(GOLDEN, fixture))
    monkeypatch.chdir(tmp_path)
    path = saveToVTK((g["ux"], g["uy"], g["uz"]), g["rho"], "ldc", "00007", (g["gx"], g["gy"], g["gz"]))
    assert path == "./ldc.00007.vtr"
    assert open(path, "rb").read() == g["vtr"].tobytes()


def test_vtr_rejects_mismatched_shapes(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    a = np.zeros((4, 3, 1))
    with pytest.raises(ValueError):
        saveToVTK((a, a, a), np.zeros((4, 4, 1)), "p", "0", (np.arange(4.), np.arange(3.), np.arange(1.)))


def test_vtr_correct_mode_is_point_data_of_the_same_grid(tmp_path, monkeypatch):
    """correct=True: the reference attaches X*Y values as CELL data to a grid that has (X-1)*(Y-1) cells
    (MRT.py:91-94,606-610 vs pyevtk/hl.py:152-153); the correct mode writes the same bytes as POINT data (X*Y points)."""
    g = np.load(os.path.join(GOLDEN, "vtr_4x3.npz"))
    monkeypatch.chdir(tmp_path)
    ref = g["vtr"].tobytes()
    path = saveToVTK((g["ux"], g["uy"], g["uz"]), g["rho"], "ldc", "00001", (g["gx"], g["gy"], g["gz"]), correct=True)
    got = open(path, "rb").read()
    assert got != ref and len(got) == len(ref) + 2 * (len("PointData") - len("CellData"))
    assert got.replace(b"PointData", b"CellData") == ref            # same extents, offsets and appended blocks
    nx, ny = g["ux"].shape[:2]
    assert b'WholeExtent="0 %d 0 %d 0 0"' % (nx - 1, ny - 1) in got   # nx * ny points = one per value
    with pytest.raises(ValueError):                                  # a value count that does not match the points
        saveToVTK((g["ux"][:-1], g["uy"][:-1], g["uz"][:-1]), g["rho"][:-1], "ldc", "2", (g["gx"], g["gy"], g["gz"]), correct=True)


def test_locate_vortices_finds_two_minima_away_from_the_walls():
    """MRT_GPU.py:764-778: the first minimum of |u|^2 outside a wall margin of X/40 cells, then the next one outside a box of
    that half-width around the first; on a synthetic field with two known zeros (and a third, deeper one inside the wall margin)."""
    X = Y = 200
    off = X // 40
    xx, yy = np.meshgrid(np.arange(X), np.arange(Y), indexing="ij")
    c1, c2, cw = (120, 70), (40, 150), (2, 100)                      # cw sits inside the wall margin: must be ignored
    d = lambda c: np.sqrt((xx - c[0]) ** 2.0 + (yy - c[1]) ** 2.0)  # noqa: E731
    speed = np.minimum(np.minimum(d(c1) * 1.0, d(c2) * 1.5 + 0.01), d(cw) * 0.1) * 1e-3
    u = np.stack([speed, np.zeros_like(speed)]).astype(np.float32)
    loc1, loc2 = ghia.locate_vortices(u, 0.08)
    assert loc1 == c1 and loc2 == c2
    assert min(loc1[0], loc1[1], X - 1 - loc1[0], Y - 1 - loc1[1]) >= off
    # a second minimum right next to the first is masked by the box around it
    u2 = u.copy()
    u2[0, c1[0] + 2, c1[1] + 1] = 0.0
    assert ghia.locate_vortices(u2, 0.08)[1] == c2


def test_vortex_table_helpers_and_masked_profile_errors():
    """r03: the comparisons with GhiaData.csv's vortex table (MRT.py:105,113-116) and the masks for its known bad centreline entries.
    vortex_position is the reference's plot mapping (MRT.py:551-553); a node put exactly on a table vortex has distance ~0 to it; the
    primary-vortex search looks in the central box only; mask_typos removes exactly the listed entries."""
    X = Y = 256
    j = ghia.RE_COLUMNS.index(1000)
    # the node whose plot coordinates are the table's BL1 vortex at Re = 1000 (0.0859, 0.0781)
    loc = (int(round(ghia.VORTEX_GHIA[2, j] * X)), Y - 1 - int(round(ghia.VORTEX_GHIA[9, j] * Y)))
    d, row = ghia.nearest_vortex_error(loc, 1000, X, Y)
    assert row == 2 and d < 1.0 / X
    assert ghia.vortex_position((0, Y - 1), X, Y) == (0.0, 0.0)
    xx, yy = np.meshgrid(np.arange(X), np.arange(Y), indexing="ij")
    prim = (int(round(ghia.VORTEX_GHIA[0, j] * X)), Y - 1 - int(round(ghia.VORTEX_GHIA[7, j] * Y)))
    speed = np.minimum(np.hypot(xx - prim[0], yy - prim[1]) + 0.5, np.hypot(xx - loc[0], yy - loc[1])) * 1e-3   # the corner eddy is the deeper minimum
    u = np.stack([speed, np.zeros_like(speed)])
    assert ghia.locate_vortices(u, 0.08)[0] == loc                     # the reference's search finds the corner eddy first ...
    dx, dy = ghia.primary_vortex_error(u, 1000, 0.08)                  # ... the central-box search the primary vortex
    assert abs(dx) < 1.0 / X and abs(dy) < 1.0 / Y
    # masks: a field that reproduces the Re = 400 table exactly except where the table is wrong
    n = 257
    Yg, Uxg, Xg, Uyg = ghia.ghia_profiles(400)
    good = Uyg.copy(); good[2] = -0.15663; good[5] = -0.33827
    u = np.zeros((2, n, n))
    u[1, :, n // 2] = np.interp(np.arange(n) / (n - 1.0), Xg[::-1], good[::-1]) * 0.08
    u[0, n // 2, :] = np.interp(1.0 - np.arange(n) / (n - 1.0), Yg[::-1], Uxg[::-1]) * 0.08
    ex, ey = ghia.profile_errors(u, 400, 0.08)
    mx, my = ghia.profile_errors(u, 400, 0.08, mask_typos=True)
    assert ey > 0.3 and my < 0.02 and ex < 0.02 and mx == ex
    assert ("uy", 400, 2) in ghia.TYPOS and ("uy", 400, 5) in ghia.PAPER_MISPRINTS


def test_partition_rows():
    assert partition_rows(4096, 1) == [(0, 4096)]
    assert partition_rows(8192, 8) == [(i * 1024, 1024) for i in range(8)]
    p = partition_rows(103, 4)
    assert [n for _, n in p] == [26, 26, 26, 25] and p[0][0] == 0 and p[-1][0] + p[-1][1] == 103
    for (a, n), (b, _) in zip(p[:-1], p[1:]):
        assert a + n == b
    with pytest.raises(ValueError):
        partition_rows(7, 4)
    assert neighbours(0, 4) == (None, 1) and neighbours(3, 4) == (2, None) and neighbours(0, 1) == (None, None)


def test_tuning_switches_map_to_lbm_params_flags():
    """CavitySolver(tuning={...}) -> (tb_steps, frame_seg, flags) of lbm_params: every switch sets exactly its bit, defaults set none,
    unknown keys are refused (the library reads no environment: this is the only way in)."""
    from latticeboltzmannsimulations_amd import _lib as L
    from latticeboltzmannsimulations_amd.solver import _tuning
    assert _tuning(None) == (0, 0, 0) and _tuning({}) == (0, 0, 0)
    assert _tuning(dict(tb_steps=4, frame_seg=32)) == (4, 32, 0)
    off = dict(deep_halo=L.LBM_FLAG_NO_DEEP_HALO, frame_fused=L.LBM_FLAG_FRAME_UNFUSED, frame_lds=L.LBM_FLAG_NO_FRAME_LDS,
               comm_priority=L.LBM_FLAG_COMM_PRIORITY_OFF, frame_wide=L.LBM_FLAG_FRAME_NARROW, edge_first=L.LBM_FLAG_NO_EDGE_FIRST,
               edge_reserve=L.LBM_FLAG_NO_EDGE_RESERVE, xcd_bands=L.LBM_FLAG_NO_XCD_BANDS, tail_tiles=L.LBM_FLAG_NO_TAIL_TILES)
    for key, bit in off.items():
        assert _tuning({key: False}) == (0, 0, bit) and _tuning({key: True}) == (0, 0, 0), key
    assert _tuning(dict(eager_lag=True))[2] == L.LBM_FLAG_EAGER_LAG and _tuning(dict(frame_fused_batch=True))[2] == L.LBM_FLAG_FRAME_FUSED_BATCH
    assert _tuning(dict(nt=True))[2] == L.LBM_FLAG_NT_ON and _tuning(dict(nt=False))[2] == L.LBM_FLAG_NT_OFF
    assert _tuning(dict(frame_beside=True))[2] == L.LBM_FLAG_FRAME_BESIDE_ON and _tuning(dict(frame_beside=False))[2] == L.LBM_FLAG_FRAME_BESIDE_OFF
    bits = [getattr(L, n) for n in dir(L) if n.startswith("LBM_FLAG_")]
    assert len(set(bits)) == len(bits) and all(b & (b - 1) == 0 for b in bits)       # distinct single bits
    with pytest.raises(ValueError, match="unknown tuning"):
        _tuning(dict(no_such_switch=True))


def test_global_mean_u_of_one_rank_is_the_row_weighted_mean():
    """slab.global_mean_u without a process group: the slab's mean times its share of the rows (what the all-reduce sums)."""
    from latticeboltzmannsimulations_amd.slab import global_mean_u

    class Fake:
        ny_local, ny = 25, 100

        def mean_u(self):
            return 0.04
    assert abs(global_mean_u(Fake()) - 0.01) < 1e-15
