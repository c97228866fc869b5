"""Drop-in front end for the reference's GPU script (MRT_GPU.py): the same knobs in, the same
``u[2, X, Y]`` / ``rho[X, Y]`` host arrays, stdout lines, ``./output/ldc_#####.png`` and
``./output/ldc.#####.vtr`` out -- with the per-step update running in liblbm_hip.so.

The reference is a flat script configured by editing module constants (MRT_GPU.py:38-58);
here they are keyword arguments of :func:`run_cavity` with the reference's names and
defaults, and ``python -m latticeboltzmannsimulations_amd.mrt_gpu --help`` exposes them.

Differences, on purpose:
* steps between two output iterations are enqueued in one ``lbm_step(n)`` call instead of one
  Python-level launch pair per step (MRT_GPU.py:707-732); the state at every output iteration
  is identical.
* ``SaveVTK=True`` works (in the reference the import is commented out, MRT.py:18, and the
  call raises NameError).
* ``turb=1`` (the reference default): the Smagorinsky closure of MRT_GPU.py:368-387 with its
  effective constant Cs2 = 0.025; the Van Driest damping lines are dead code in the reference
  (overwritten at line 374) and are not reproduced.
"""
import os
from timeit import default_timer as timer

import numpy as np

from . import ghia
from .VTKWrapper import saveToVTK
from .solver import CavitySolver


class CavityResult:
    def __init__(self):
        self.u = None
        self.rho = None
        self.iterations = 0
        self.converged = False
        self.regression = []   # (iteration, value) at every output iteration
        self.elapsed = 0.0
        self.mlups = 0.0


CS2_EFFECTIVE, CS_BULK = 0.025, 0.16     # MRT_GPU.py:350,374-376: Van Driest damping is overwritten by Cs2 = 0.025


def _dashboard(path, u, rho, It, hist, Re, RT, regime, BC, xsize, ysize, uLB, relax, tau_mean=None):
    """PNG dashboard of MRT_GPU.py:786-870 (centreline plots vs Ghia, streamlines with vortex
    markers, regression history, parameter text)."""
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import pyplot

    Yg, Uxg, Xg, Uyg = ghia.ghia_profiles(Re)
    Xv, Yv = ghia.ghia_vortices(Re)
    YNorm = np.arange(ysize, 0, -1, dtype="float64") / ysize
    XNorm = np.arange(0, xsize, 1, dtype="float64") / xsize
    loc1, loc2 = ghia.locate_vortices(u, uLB)
    f = pyplot.figure(figsize=(30, 16))
    s1 = pyplot.subplot2grid((2, 15), (0, 0), colspan=4, rowspan=1)
    s2 = pyplot.subplot2grid((2, 15), (0, 5), colspan=4, rowspan=1)
    s3 = pyplot.subplot2grid((2, 15), (0, 10), colspan=5, rowspan=1)
    s4 = pyplot.subplot2grid((2, 15), (1, 0), colspan=10, rowspan=1)
    Ux, Uy = ghia.centrelines(u, uLB)
    s1.plot(Ux, YNorm, label="LBM"); s1.plot(Uxg, Yg, "g*", label="Ghia")
    s1.set_title("Ux on middle column", fontsize=20, y=1.02); s1.legend(loc="center right")
    s1.set_xlabel("Ux", fontsize=20); s1.set_ylabel("Y-position", fontsize=20)
    s2.plot(XNorm, Uy, label="LBM"); s2.plot(Xg, Uyg, "g*", label="Ghia")
    s2.set_title("Uy on middle row", fontsize=20, y=1.02); s2.legend(loc="upper right")
    s2.set_xlabel("X-position", fontsize=20); s2.set_ylabel("Uy", fontsize=20)
    color1 = (np.sqrt(u[0] ** 2 + u[1] ** 2) / uLB).transpose()
    try:
        # YNorm decreases (y = 0 is the lid): streamplot wants increasing coordinates
        strm = s3.streamplot(XNorm, YNorm[::-1], u[0].transpose()[::-1], u[1].transpose()[::-1],
                             color=color1[::-1], cmap=pyplot.cm.jet)
        pyplot.colorbar(strm.lines, ax=s3)
    except Exception:
        s3.imshow(color1, extent=(0, 1, 0, 1), cmap=pyplot.cm.jet)
    s3.plot(loc1[0] / xsize, (ysize - 1 - loc1[1]) / ysize, "ro", label="Vortex1")
    s3.plot(loc2[0] / xsize, (ysize - 1 - loc2[1]) / ysize, "mo", label="Vortex2")
    s3.plot(Xv, Yv, "ks", label="Ghia")
    s3.set_title("Velocity Streamlines - LBM", fontsize=20, y=1.02)
    s3.set_xlabel("X-position", fontsize=20); s3.set_ylabel("Y-position", fontsize=20)
    s4.plot([h[0] for h in hist], [h[1] for h in hist])
    s4.set_title("Regression value - Ux_MiddleColumn")
    s4.set_xlabel("time iteration", fontsize=20); s4.set_ylabel("Regression value", fontsize=20)
    pyplot.figtext(0.5, 0.3, "Current Regression value is")
    pyplot.figtext(0.5, 0.28, str(round(hist[-1][1], 4)))
    pyplot.figtext(0.65, 0.45, "Square dots in above figure represent vortex locations from Ghia data")       # MRT_GPU.py:846-847
    pyplot.figtext(0.65, 0.43, "Circular dots represent vortex locations of current simulation")
    pyplot.figtext(0.65, 0.35, "LBM parameters: " + RT, fontsize=20)
    pyplot.figtext(0.65, 0.31, "Grid size: " + str(xsize) + "*" + str(ysize))
    pyplot.figtext(0.65, 0.29, "Re: " + str(Re) + "    " + "BoundaryCondition: " + BC)
    pyplot.figtext(0.65, 0.27, "Lid velocity in LB units: " + str(uLB) + "    dx* and dt* hardcoded as 1")
    pyplot.figtext(0.65, 0.25, "tau - related to dynamic viscosity: " + str(round(1.0 / relax["omega"], 3)))
    if RT == "SRT":
        pyplot.figtext(0.65, 0.23, "omega: " + str(round(relax["omega"], 2)))
    elif RT == "TRT":
        pyplot.figtext(0.65, 0.23, "omega_plus, omega_minus, delta: " + str(round(relax["omega"], 3)) + " , "
                       + str(round(relax["omegam"], 3)) + " , " + str(round(1.0 / 3.5, 3)))
    else:
        pyplot.figtext(0.65, 0.23, "omega_nu, omega_e, omega_eps, omega_q: ")
        pyplot.figtext(0.65, 0.21, " , ".join(str(round(relax[k], 3)) for k in ("omega", "omega_e", "omega_eps", "omega_q")))
    if tau_mean is not None:
        # MRT_GPU.py:862-866.  The reference prints these two lines under its `regime == 'Turbulent'` label (which it assigns to
        # turb = 0, lines 277-280) from host names that do not exist (`Cs`, `Csbulk`: NameError) and from a `tauS` that is never
        # downloaded; here they appear when the closure is on, with the constant the kernel really uses and the mean of taus_g.
        pyplot.figtext(0.65, 0.17, "Smagorinsky constant, Cs = " + str(round(CS2_EFFECTIVE ** 0.5, 4)) + " at wall to " + str(CS_BULK) + " at bulk")
        pyplot.figtext(0.65, 0.15, "Mean relaxation time, tau+tau_turbulent, is  " + str(tau_mean))
    f.suptitle("Lid Driven Cavity - Re" + str(int(Re)) + " " + regime + " " + RT + " " + BC + " " + str(xsize) + "*"
               + str(ysize), fontsize=30, y=1.04)
    pyplot.savefig(path, bbox_inches="tight", pad_inches=0.4)
    pyplot.close(f)


def run_cavity(maxIt=3000000, Re=10000.0, RT="SRT", turb=1, xsize=32 * 5, ysize=32 * 5, uLB=0.08,
               Pinterval=3000, SavePlot=True, SaveVTK=False, project="ldc", OutputFolder="./output",
               dtype=np.float32, semantics="mrt_gpu", device=0, quiet=False, solver_factory=None, arith="strict",
               convergence="host", vtk_correct=False):
    """Run the lid-driven cavity like MRT_GPU.py does; returns a :class:`CavityResult`.

    Argument names and defaults are the module constants of MRT_GPU.py:38-58.
    xsize / ysize need not be multiples of 32 here.  `solver_factory` (default: CavitySolver, i.e. liblbm_hip.so)
    exists so that this driver logic -- output iterations, metrics, files, convergence stop -- can be unit-tested
    with a stand-in stepper; it is not a fallback: nothing in this package provides another stepper.
    convergence: 'host' (default) -- the reference's test as written, on NumPy's float32 mean of the downloaded u
    (MRT_GPU.py:883-889); 'device' -- the same test on lbm_mean_u(), the mean reduced on the GPU in double (8 bytes cross
    PCIe instead of the field; the two means differ in the last bits of a float, so a run may stop one check apart).
    vtk_correct: write the .vtr files as point data (VTKWrapper.saveToVTK(correct=True)) instead of the reference's layout."""
    if convergence not in ("host", "device"):
        raise ValueError("convergence must be 'host' or 'device'")
    say = (lambda *a: None) if quiet else print
    tstart = timer()
    say("the value of uLB is ", uLB)
    say("xsize value is ", xsize)
    make = CavitySolver if solver_factory is None else solver_factory
    extra = {} if arith == "strict" else {"arith": arith}      # 'fast': see CavitySolver (agrees with 'strict' to rounding)
    solver = make(xsize, ysize, Re, RT=RT, uLB=uLB, semantics=semantics, dtype=dtype, turb=turb, device=device, **extra)
    relax = solver.relax
    say("Re chosen  is ", Re)
    say("RT chosen is ", RT)
    say("Turbulence is on" if turb == 1 else "Turbulence is off")
    say("the value of tau(/Dt) is ", 1 / relax["omega"])
    if RT == "SRT":
        say(" the value of omega is ", relax["omega"])
    elif RT == "TRT":
        say("the value of deltaTRT is ", 1.0 / 3.5)
        say("omegap, omegam :", round(relax["omega"], 4), " , ", round(relax["omegam"], 4))
    else:
        say("omega omegap omegam omega_nu omega_e omega_eps omega_q")
        say(relax["omega"], relax["omega"], relax["omegam"], relax["omega"], relax["omega_e"], relax["omega_eps"],
            relax["omega_q"])
    if (SavePlot or SaveVTK) and not os.path.isdir(OutputFolder):
        try:
            os.makedirs(OutputFolder)
        except OSError:
            pass
    grid = (np.arange(0, xsize, dtype="float64"), np.arange(0, ysize, dtype="float64"), np.arange(0, 1, dtype="float64"))
    velZ = np.zeros((xsize, ysize, 1), dtype=np.float32)   # same dtype as the float32 host fields (MRT_GPU.py:207)
    regime = "Laminar" if turb == 1 else "Turbulent"   # labels exactly as (mis)assigned at MRT_GPU.py:277-280
    BC = "EB-NEBB "
    res = CavityResult()
    u = np.zeros((2, xsize, ysize), dtype=np.float32)
    mean_past = 0.0
    count = 0
    done = 0          # iterations performed
    have_ghia = int(round(float(Re))) in ghia.RE_COLUMNS
    outputs = SaveVTK or SavePlot
    It = 0
    while It < maxIt:
        # iterations It .. next output iteration (inclusive) in one enqueue
        nxt = It if (It % Pinterval == 0) else min(maxIt - 1, (It // Pinterval + 1) * Pinterval)
        if not outputs:
            nxt = maxIt - 1
        solver.step(nxt - It + 1)
        done = nxt + 1
        It = nxt
        if (It % Pinterval == 0) and outputs:
            u_past = u.copy()
            u, rho = solver.get_fields(out_dtype=np.float32)
            say("current iteration :", It)
            if have_ghia:
                reg_val = ghia.r2_value(u, Re, uLB)
                res.regression.append((It, float(reg_val)))
                say("current regression value is " + str(reg_val))
            say("current mean velocity value is " + str(np.mean(u) / uLB))
            if SavePlot and have_ghia:
                tau_mean = float(np.mean(solver.get_tau())) if (turb == 1 and hasattr(solver, "get_tau")) else None
                _dashboard(os.path.join(OutputFolder, project + "_" + str(int(It / Pinterval)).zfill(5) + ".png"),
                           u, rho, It, res.regression, Re, RT, regime, BC, xsize, ysize, uLB, relax, tau_mean)
            if SaveVTK:
                Vel = np.reshape(u, (2, xsize, ysize, 1))
                cwd = os.getcwd()
                os.chdir(OutputFolder)
                try:
                    saveToVTK((Vel[0], Vel[1], velZ), np.reshape(rho, (xsize, ysize, 1)), project,
                              str(int(It / Pinterval)).zfill(5), grid, correct=vtk_correct)
                finally:
                    os.chdir(cwd)
            say("time elapsed is ", (timer() - tstart), "seconds")
            if convergence == "device":
                mean_now = solver.mean_u()
                hit = abs(mean_now - mean_past) / uLB < 0.00000001
                mean_past = mean_now
            else:
                hit = abs(np.mean(u) - np.mean(u_past)) / uLB < 0.00000001      # MRT_GPU.py:883-889
            if hit:
                count = count + 1
                if count > 5:
                    say("breaking out of loop because of convergence")
                    res.converged = True
                    break
            if It == maxIt - 1:
                say("max iterations reached. More needed for convergence.")
        It += 1
    solver.sync()
    res.u, res.rho = solver.get_fields(out_dtype=np.float32)
    res.iterations = done
    res.elapsed = timer() - tstart
    res.mlups = xsize * ysize * done * 1e-6 / res.elapsed        # as printed by MRTTiledPull.py:703
    say("TOTAL time elapsed is ", res.elapsed, "seconds")
    solver.close()
    return res


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="D2Q9 lid-driven cavity on MI355X (drop-in for MRT_GPU.py)")
    ap.add_argument("--maxIt", type=int, default=30000)
    ap.add_argument("--Re", type=float, default=1000.0)
    ap.add_argument("--RT", choices=["SRT", "TRT", "MRT"], default="MRT")
    ap.add_argument("--turb", type=int, default=1)
    ap.add_argument("--xsize", type=int, default=160)
    ap.add_argument("--ysize", type=int, default=160)
    ap.add_argument("--uLB", type=float, default=0.08)
    ap.add_argument("--Pinterval", type=int, default=3000)
    ap.add_argument("--no-plot", action="store_true")
    ap.add_argument("--vtk", action="store_true")
    ap.add_argument("--project", default="ldc")
    ap.add_argument("--OutputFolder", default="./output")
    ap.add_argument("--dtype", choices=["float32", "float64"], default="float32")
    ap.add_argument("--semantics", choices=["mrt_gpu", "mrt_py"], default="mrt_gpu")
    ap.add_argument("--arith", choices=["strict", "fast"], default="strict")
    ap.add_argument("--convergence", choices=["host", "device"], default="host")
    ap.add_argument("--vtk-correct", action="store_true", help="write .vtr point data (consistent file) instead of the reference's layout")
    a = ap.parse_args(argv)
    r = run_cavity(maxIt=a.maxIt, Re=a.Re, RT=a.RT, turb=a.turb, xsize=a.xsize, ysize=a.ysize, uLB=a.uLB,
                   Pinterval=a.Pinterval, SavePlot=not a.no_plot, SaveVTK=a.vtk, project=a.project,
                   OutputFolder=a.OutputFolder, dtype=np.dtype(a.dtype), semantics=a.semantics, arith=a.arith,
                   convergence=a.convergence, vtk_correct=a.vtk_correct)
    print("MLUPS : ", r.mlups)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
