"""Ghia, Ghia & Shin (1982) lid-driven-cavity reference table and the validation metrics of the
reference scripts (output stage).

The numbers are DATA: the table the reference ships as GhiaData.csv (loaded at MRT.py:104-116),
embedded so that the front end prints the same regression values as the reference.  They are kept
exactly as in that file, including its transcription slips versus the paper (+0.15663 in the Re=400
Uy column, -0.86636 in the Re=3200 Ux column, +0.03111 in the Re=10000 Ux column).
tests/test_host_logic.py checks this module against tests/golden/ghia.npz."""
import numpy as np

RE_COLUMNS = (100, 400, 1000, 3200, 5000, 7500, 10000)   # Re_dict, MRT.py:109
Y_GHIA = np.array([1.0, 0.9766, 0.9688, 0.9609, 0.9531, 0.8516, 0.7344, 0.6172, 0.5, 0.4531, 0.2831, 0.1719, 0.1016, 0.0703, 0.0625, 0.0547, 0.0])   # middle column sample heights (1 = lid)
X_GHIA = np.array([1.0, 0.9688, 0.9609, 0.9531, 0.9453, 0.9063, 0.8594, 0.8047, 0.5, 0.2344, 0.2266, 0.1563, 0.0938, 0.0781, 0.0703, 0.0625, 0.0])   # middle row sample abscissae
UX_GHIA = np.array([
    [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0],
    [0.84123, 0.75837, 0.65928, 0.53236, 0.48223, 0.47224, 0.47221],
    [0.78871, 0.68439, 0.57492, 0.48296, 0.4612, 0.47048, 0.47783],
    [0.73722, 0.61756, 0.51117, 0.46547, 0.45992, 0.47323, 0.4807],
    [0.68717, 0.55892, 0.46604, 0.46101, 0.46036, 0.47167, 0.47804],
    [0.23151, 0.29093, 0.33304, 0.34682, 0.33556, 0.34228, 0.34635],
    [0.00332, 0.16256, 0.18719, 0.19791, 0.20087, 0.20591, 0.20673],
    [-0.13641, 0.02135, 0.05702, 0.07156, 0.08183, 0.08342, 0.08344],
    [-0.20581, -0.11477, -0.0608, -0.04272, -0.03039, -0.038, 0.03111],
    [-0.2109, -0.17119, -0.10648, -0.86636, -0.07404, -0.07503, -0.0754],
    [-0.15662, -0.32726, -0.27805, -0.24427, -0.22855, -0.23176, -0.23186],
    [-0.1015, -0.24299, -0.38289, -0.34323, -0.3305, -0.32393, -0.32709],
    [-0.06434, -0.14612, -0.2973, -0.41933, -0.40435, -0.38324, -0.38],
    [-0.04775, -0.10338, -0.2222, -0.37827, -0.43643, -0.43025, -0.41657],
    [-0.04192, -0.09266, -0.20196, -0.35344, -0.42901, -0.4359, -0.42537],
    [-0.03717, -0.08186, -0.18109, -0.32407, -0.41165, -0.43154, -0.42735],
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
])   # [17 points, 7 Reynolds numbers]
UY_GHIA = np.array([
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
    [-0.05906, -0.12146, -0.21388, -0.39017, -0.49774, -0.53858, -0.54302],
    [-0.07391, 0.15663, -0.27669, -0.47425, -0.55069, -0.55216, -0.52987],
    [-0.08864, -0.19254, -0.33714, -0.52357, -0.55408, -0.52347, -0.49099],
    [-0.10313, -0.22847, -0.39188, -0.54053, -0.52876, -0.4859, -0.45863],
    [-0.16914, -0.23827, -0.5155, -0.44307, -0.41442, -0.4105, -0.41496],
    [-0.22445, -0.44993, -0.42665, -0.37401, -0.36214, -0.36213, -0.36737],
    [-0.24533, -0.38598, -0.31966, -0.31184, -0.30018, -0.30448, -0.30719],
    [0.05454, 0.05186, 0.02526, 0.00999, 0.00945, 0.00824, 0.00831],
    [0.17527, 0.30174, 0.32235, 0.28188, 0.2728, 0.27348, 0.27224],
    [0.17507, 0.30203, 0.33075, 0.2903, 0.28066, 0.28117, 0.28003],
    [0.16077, 0.28124, 0.37095, 0.37199, 0.35368, 0.3506, 0.3507],
    [0.12317, 0.22965, 0.32627, 0.42768, 0.42951, 0.41824, 0.41487],
    [0.1089, 0.2092, 0.30353, 0.41906, 0.43648, 0.43654, 0.43124],
    [0.10091, 0.19713, 0.29012, 0.40917, 0.43329, 0.4403, 0.43733],
    [0.09233, 0.1836, 0.27485, 0.3956, 0.42447, 0.43979, 0.43983],
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
])
VORTEX_GHIA = np.array([   # rows: X of Primary, Top, BL1, BR1, BL2, BR2, BR3, then Y of the same (MRT.py:105)
    [0.6172, 0.5547, 0.5313, 0.5165, 0.5117, 0.5117, 0.5117],
    [0.0, 0.0, 0.0, 0.0547, 0.0625, 0.0664, 0.0703],
    [0.0313, 0.0508, 0.0859, 0.0859, 0.0703, 0.0645, 0.0586],
    [0.9453, 0.8906, 0.8594, 0.8125, 0.8086, 0.7813, 0.7656],
    [0.0, 0.0039, 0.0, 0.0078, 0.0117, 0.0117, 0.0156],
    [0.0, 0.9922, 0.9922, 0.9844, 0.9805, 0.9492, 0.9336],
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.9961, 0.9961],
    [0.7344, 0.6055, 0.5625, 0.5469, 0.5352, 0.5322, 0.5333],
    [0.0, 0.0, 0.0, 0.8984, 0.9102, 0.9141, 0.9141],
    [0.0391, 0.0469, 0.0781, 0.1094, 0.1367, 0.1504, 0.1641],
    [0.0625, 0.125, 0.1094, 0.0859, 0.0742, 0.0625, 0.0586],
    [0.0, 0.0039, 0.0, 0.0078, 0.0078, 0.0117, 0.0195],
    [0.0, 0.0078, 0.0078, 0.0078, 0.0195, 0.043, 0.0625],
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0039, 0.0039],
])


def _col(Re):
    Re = int(round(float(Re)))
    if Re not in RE_COLUMNS:
        raise KeyError(f"Ghia table has no column for Re={Re}; available: {RE_COLUMNS}")
    return RE_COLUMNS.index(Re)


def ghia_profiles(Re):
    """(Y, Ux(Y) on the middle column, X, Uy(X) on the middle row) -- MRT.py:106-111."""
    j = _col(Re)
    return Y_GHIA, UX_GHIA[:, j], X_GHIA, UY_GHIA[:, j]


def ghia_vortices(Re):
    """Non-zero vortex centre coordinates (X_Vor, Y_Vor) -- MRT.py:113-116."""
    j = _col(Re)
    xv, yv = VORTEX_GHIA[0:7, j], VORTEX_GHIA[7:14, j]
    return xv[xv != 0], yv[yv != 0]


def sample_rows(ysize):
    """LBMy: lattice rows the reference samples the middle column at (MRT.py:119-120):
    int(Y_Ghia * (ysize-1)) without the first (lid) point.  Note the reference then REVERSES
    the sampled vector before comparing (MRT.py:559), an approximate mapping kept as is."""
    return np.array(Y_GHIA * (ysize - 1)).astype(int)[1:]


def centrelines(u, uLB):
    """Ux on the middle column and Uy on the middle row, in lid-velocity units (MRT.py:533,540)."""
    _, X, Y = u.shape
    return u[0, int(X / 2), :] / uLB, u[1, :, int(Y / 2)] / uLB


def regression_value(u, Re, uLB):
    """The CPU script's home-made agreement number (MRT.py:557-561)."""
    _, X, Y = u.shape
    ux_ghia = ghia_profiles(Re)[1]
    ux_tmp = u[0, int(X / 2), sample_rows(Y)] / uLB
    flipped = np.fliplr(np.atleast_2d(ux_tmp))[0]
    tmp = abs((abs(ux_ghia[:-1]) - abs(flipped)) / (len(ux_tmp) * np.maximum(abs(ux_ghia[:-1]), abs(flipped))))
    return 1 - np.sum(tmp, axis=0)


def r2_value(u, Re, uLB):
    """The GPU script's metric (MRT_GPU.py:815-821): coefficient of determination
    r2_score(Ux_Ghia[:-1], flipped(Ux_LBM + 0.001)), written out (no sklearn dependency)."""
    _, X, Y = u.shape
    y_true = ghia_profiles(Re)[1][:-1]
    ux_tmp = u[0, int(X / 2), sample_rows(Y)] / uLB
    ux_tmp = ux_tmp + 0.001
    y_pred = np.fliplr(np.atleast_2d(ux_tmp))[0]
    ss_res = np.sum((y_true - y_pred) ** 2)
    ss_tot = np.sum((y_true - np.mean(y_true)) ** 2)
    return 1.0 - ss_res / ss_tot


def locate_vortices(u, uLB):
    """Two minima of |u|^2 away from the walls (MRT_GPU.py:764-778) -> ((x1, y1), (x2, y2))."""
    _, X, Y = u.shape
    usq = (u[0].astype(np.float64) ** 2 + u[1].astype(np.float64) ** 2) / (uLB ** 2)
    off = int(X / 40)
    usq[0:off, :] = np.nan
    usq[:, 0:off] = np.nan
    usq[X - 1 - off:X, :] = np.nan
    usq[:, Y - 1 - off:Y] = np.nan
    loc1 = np.unravel_index(np.nanargmin(usq), usq.shape)
    usq[max(loc1[0] - off, 0):loc1[0] + off, max(loc1[1] - off, 0):loc1[1] + off] = np.nan
    loc2 = np.unravel_index(np.nanargmin(usq), usq.shape)
    return tuple(int(v) for v in loc1), tuple(int(v) for v in loc2)


# The three transcription slips of GhiaData.csv versus the paper (header of this module): (column, Re, row).  The table is kept as the
# reference ships it; a comparison that wants the physics leaves these points out (`mask_typos`).
TYPOS = (("uy", 400, 2), ("ux", 3200, 9), ("ux", 10000, 8))
# One more entry no solution can meet: Uy(x = 0.9063) at Re = 400 reads -0.23827 in the CSV exactly as in the paper's table, between
# -0.22847 and -0.44993; every converged run here (oracle and HIP, 128^2 .. 256^2) gives -0.34 .. -0.35 there and is within 0.03 of all
# its neighbours -- a misprint of the paper itself (commonly read as -0.33827).  Masked together with the transcription slips.
PAPER_MISPRINTS = (("uy", 400, 5),)


def profile_errors(u, Re, uLB, mask_typos=False):
    """Geometrically consistent comparison with Ghia (not in the reference): the lattice node
    (x, y) sits at (x/(X-1), 1 - y/(Y-1)); linear interpolation of the LBM centrelines at the
    Ghia sample positions.  Returns (max |dUx|, max |dUy|) in lid-velocity units; mask_typos leaves
    out the table entries listed in TYPOS and PAPER_MISPRINTS."""
    _, X, Y = u.shape
    Yg, Uxg, Xg, Uyg = ghia_profiles(Re)
    ux, uy = centrelines(u.astype(np.float64), uLB)
    height = 1.0 - np.arange(Y) / (Y - 1.0)            # decreasing in y
    ux_i = np.interp(Yg, height[::-1], ux[::-1])
    # cy = +1 populations move towards y - 1, i.e. towards the lid: u[1] > 0 is Ghia's v > 0
    uy_i = np.interp(Xg, np.arange(X) / (X - 1.0), uy)
    kx, ky = np.ones(len(Yg), bool), np.ones(len(Xg), bool)
    if mask_typos:
        for col, re_, row in TYPOS + PAPER_MISPRINTS:
            if re_ == int(round(float(Re))):
                (kx if col == "ux" else ky)[row] = False
    return float(np.max(np.abs(ux_i - Uxg)[kx])), float(np.max(np.abs(uy_i - Uyg)[ky]))


def vortex_position(loc, X, Y):
    """A located lattice node in the coordinates the reference plots it in beside Ghia's vortex table (MRT.py:551-553,
    MRT_GPU.py:803-805): (i / xsize, (ysize_max - j) / ysize) -- x from the left wall, y from the BOTTOM wall, as in the table."""
    return loc[0] / float(X), (Y - 1 - loc[1]) / float(Y)


def nearest_vortex_error(loc, Re, X, Y):
    """Distance from a located node (locate_vortices returns the two smallest minima of |u|^2 -- WHICH vortices they are depends on the
    flow: at Re >= 400 the first is the bottom-left corner eddy, not the primary vortex) to the nearest entry of Ghia's vortex table
    for this Re, in the reference's plot coordinates: (distance, index of that table row: 0 Primary, 1 Top, 2 BL1, 3 BR1, ...)."""
    px, py = vortex_position(loc, X, Y)
    j = _col(Re)
    xv, yv = VORTEX_GHIA[0:7, j], VORTEX_GHIA[7:14, j]
    d = np.where((xv != 0) | (yv != 0), np.hypot(xv - px, yv - py), np.inf)
    i = int(np.argmin(d))
    return float(d[i]), i


def primary_vortex_error(u, Re, uLB):
    """(dx, dy) between the minimum of |u|^2 inside the central box 0.25 .. 0.75 of the cavity -- the primary vortex; not in the
    reference, whose two-minima search does not tell the vortices apart -- and the `Primary` row of Ghia's vortex table (VORTEX_GHIA
    rows 0 / 7; MRT.py:105,113-116), in the reference's plot coordinates."""
    _, X, Y = u.shape
    usq = u[0].astype(np.float64) ** 2 + u[1].astype(np.float64) ** 2
    box = np.full_like(usq, np.inf)
    x0, x1, y0, y1 = X // 4, X - X // 4, Y // 4, Y - Y // 4
    box[x0:x1, y0:y1] = usq[x0:x1, y0:y1]
    loc = np.unravel_index(np.argmin(box), box.shape)
    px, py = vortex_position(loc, X, Y)
    j = _col(Re)
    return px - VORTEX_GHIA[0, j], py - VORTEX_GHIA[7, j]
