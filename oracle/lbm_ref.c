/* CPU oracle (plain C) for the D2Q9 lid-driven-cavity step.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (latticeboltzmannsimulations_amd/) never links or calls it.
 *
 * Second, independently written restatement (the first is oracle/lbm_numpy.py) of
 *   semantics 0 "mrt_py"  : /root/reference/MRT.py:286-453
 *   semantics 1 "mrt_gpu" : /root/reference/MRT_GPU.py:336-699 (funRT SRT/TRT/MRT + funBC)
 * in the reference's host layout fin[k][x][y], y fastest, y = 0 the moving lid.
 *
 * PARITY STATUS: parity unpinned at bit level (see oracle/README.md): the reference's
 * MRT.py is not importable in the build image (numba, numexpr absent) and the reference
 * holds no golden vectors for this path.  Pinned by GhiaData.csv at physics level, by
 * operator identities, and by bit-equality with oracle/lbm_numpy.py.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: no FMA contraction, so every
 * + - * / is one IEEE operation in the order written).
 * Compiled twice: -DREAL=double -DSUF=f64 and -DREAL=float -DSUF=f32.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#define SUF f64
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* a1: MRT.py:138-140 */
static const int CX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
static const int CY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};

/* MRT.py:163-183 / MRT_GPU.py:593-612 */
static const int MGS[9][9] = {
    {1, 1, 1, 1, 1, 1, 1, 1, 1},   {-4, -1, -1, -1, -1, 2, 2, 2, 2}, {4, -2, -2, -2, -2, 1, 1, 1, 1},
    {0, 1, 0, -1, 0, 1, -1, -1, 1}, {0, -2, 0, 2, 0, 1, -1, -1, 1},   {0, 0, 1, 0, -1, 1, 1, -1, -1},
    {0, 0, -2, 0, 2, 1, 1, -1, -1}, {0, 1, -1, 1, -1, 0, 0, 0, 0},    {0, 0, 0, 0, 0, 1, -1, 1, -1}};
static const double MINV[9][9] = {
    {1.0 / 9, -1.0 / 9, 1.0 / 9, 0, 0, 0, 0, 0, 0},
    {1.0 / 9, -1.0 / 36, -1.0 / 18, 1.0 / 6, -1.0 / 6, 0, 0, 1.0 / 4, 0},
    {1.0 / 9, -1.0 / 36, -1.0 / 18, 0, 0, 1.0 / 6, -1.0 / 6, -1.0 / 4, 0},
    {1.0 / 9, -1.0 / 36, -1.0 / 18, -1.0 / 6, 1.0 / 6, 0, 0, 1.0 / 4, 0},
    {1.0 / 9, -1.0 / 36, -1.0 / 18, 0, 0, -1.0 / 6, 1.0 / 6, -1.0 / 4, 0},
    {1.0 / 9, 1.0 / 18, 1.0 / 36, 1.0 / 6, 1.0 / 12, 1.0 / 6, 1.0 / 12, 0, 1.0 / 4},
    {1.0 / 9, 1.0 / 18, 1.0 / 36, -1.0 / 6, -1.0 / 12, 1.0 / 6, 1.0 / 12, 0, -1.0 / 4},
    {1.0 / 9, 1.0 / 18, 1.0 / 36, -1.0 / 6, -1.0 / 12, -1.0 / 6, -1.0 / 12, 0, 1.0 / 4},
    {1.0 / 9, 1.0 / 18, 1.0 / 36, 1.0 / 6, 1.0 / 12, -1.0 / 6, -1.0 / 12, 0, -1.0 / 4}};

enum { SEM_MRT_PY = 0, SEM_MRT_GPU = 1 };
enum { COLL_SRT = 0, COLL_TRT = 1, COLL_MRT = 2 };

/* PROMOTE (mrt_gpu semantics, float build): the CUDA text of MRT_GPU.py mixes `double` literals into `float` expressions
 * (MRT_GPU.py:385, 410, 638-642, 652).  By C's usual arithmetic conversions every operation that has such a literal (or a
 * value already promoted by one) as an operand is a double operation, and the sub-expression is rounded to float ONCE, at
 * the assignment.  promote != 0 evaluates exactly those sub-expressions that way; operations between two floats (or an int
 * and a float) stay float operations.  In the double build the casts are no-ops: promote changes nothing (tests assert it).
 * What this still is NOT: nvcc's default --fmad=true may fuse any float multiply-add of the text into one FMA; which ones
 * is the compiler's choice and cannot be read from the text, so no restatement can claim the bits of a real run. */

/* a3: MRT.py:213-231; promote: MRT_GPU.py:410,652 `rho_l*t_g[k]*(1. + 3.0*cu + 9*0.5*cu*cu - 3.0*0.5*usqr)` --
 * rho_l*t_g[k] is float * float, the bracket is double throughout, their product double, one rounding at the store */
static inline void equ_cell(REAL rho, REAL ux, REAL uy, const REAL t[9], REAL feq[9], int promote) {
    REAL usqr = ux * ux + uy * uy;
    for (int k = 0; k < 9; ++k) {
        REAL cu = (REAL)CX[k] * ux + (REAL)CY[k] * uy;
        if (promote) {
            const REAL rt = rho * t[k];
            feq[k] = (REAL)((double)rt * (((1. + 3.0 * (double)cu) + (4.5 * (double)cu) * (double)cu) - 1.5 * (double)usqr));
        } else {
            feq[k] = (rho * t[k]) * ((((REAL)1. + (REAL)3.0 * cu) + ((REAL)4.5 * cu) * cu) - (REAL)1.5 * usqr);
        }
    }
}

/* relax[5] = omega, omegam, omega_e, omega_eps, omega_q */
#if REAL_IS_FLOAT
#define RSQRT(x) sqrtf(x)
#define RABS(x) fabsf(x)
#else
#define RSQRT(x) sqrt(x)
#define RABS(x) fabs(x)
#endif

/* Smagorinsky relaxation rate, MRT_GPU.py:368-387 (lines 370-373 are dead code: Cs2 is overwritten
 * at 374).  feq_prev / rho_prev: what funRT wrote in the previous step at this cell. */
static inline REAL smagorinsky_omega(const REAL f[9], const REAL fe_prev[9], REAL rho_prev, REAL omega, int promote) {
    const REAL tau0 = (REAL)1.0 / omega;   /* (`1.0/omega` is a double division rounded to float = the float division: one operation) */
    const REAL p1 = -f[8] + (f[7] + (-f[6] + f[5]));
    const REAL p2 = -fe_prev[8] + (fe_prev[7] + (-fe_prev[6] + fe_prev[5]));
    const REAL q = p1 - p2;
    if (promote) {
        /* MRT_GPU.py:385 `0.5*(tau + sqrt( (tau*tau + ( 18*1.4142*Cs2*abs(Qmf) )/rho_g[i] ) ) )`: tau*tau is float * float, Cs2 is the
         * float 0.025f, abs(float) is float; everything else is double (18*1.4142 is a double constant), one rounding at `tau =` */
        const REAL tt = tau0 * tau0, cs2 = (REAL)0.025, aq = RABS(q);
        const REAL tau = (REAL)(0.5 * ((double)tau0 + sqrt((double)tt + (((18 * 1.4142) * (double)cs2) * (double)aq) / (double)rho_prev)));
        return (REAL)1.0 / tau;
    }
    const REAL tau = (REAL)0.5 * (tau0 + RSQRT(tau0 * tau0 + (((REAL)(18 * 1.4142) * (REAL)0.025) * RABS(q)) / rho_prev));
    return (REAL)1.0 / tau;
}

static inline void collide_cell(int coll, const REAL f[9], REAL rho, const REAL feq[9], const REAL w[5],
                                const REAL mi[9][9], REAL out[9], int promote) {
    if (coll == COLL_SRT) { /* MRT.py:396 */
        for (int k = 0; k < 9; ++k) out[k] = f[k] - w[0] * (f[k] - feq[k]);
    } else if (coll == COLL_TRT) { /* MRT_GPU.py:455-462,514-525 */
        static const int pa[4] = {2, 5, 6, 1}, pb[4] = {4, 7, 8, 3};
        REAL fp[9], fm[9], ep[9], em[9];
        for (int i = 0; i < 4; ++i) {
            int a = pa[i], b = pb[i];
            fp[a] = (REAL)0.5 * (f[a] + f[b]); fp[b] = fp[a];
            fm[a] = (REAL)0.5 * (f[a] - f[b]); fm[b] = -fm[a];
            ep[a] = (REAL)0.5 * (feq[a] + feq[b]); ep[b] = ep[a];
            em[a] = (REAL)0.5 * (feq[a] - feq[b]); em[b] = -em[a];
        }
        fp[0] = f[0]; fm[0] = 0; ep[0] = feq[0]; em[0] = 0;
        for (int k = 0; k < 9; ++k) out[k] = (f[k] - w[0] * (fp[k] - ep[k])) - w[1] * (fm[k] - em[k]);
    } else { /* MRT_GPU.py:633-655 */
        REAL m[9], meq[9];
        const REAL wv[9] = {0, w[2], w[3], 0, w[4], 0, w[4], w[0], w[0]};
        for (int k = 0; k < 9; ++k) {
            REAL acc = (REAL)MGS[k][0] * f[0];
            for (int j = 1; j < 9; ++j) acc = acc + (REAL)MGS[k][j] * f[j];
            m[k] = acc;
        }
        REAL jx = m[3], jy = m[5];
        meq[0] = rho;
        meq[1] = (REAL)-2.0 * rho + (REAL)3.0 * (jx * jx + jy * jy);
        meq[2] = ((REAL)-3.0 * (jx * jx + jy * jy) + rho) + (REAL)9.0 * (((jx * jx) * jy) * jy);
        meq[3] = m[3];
        meq[4] = -jx + (REAL)3.0 * ((jx * jx) * jx);
        meq[5] = m[5];
        meq[6] = -jy + (REAL)3.0 * ((jy * jy) * jy);
        meq[7] = jx * jx - jy * jy;
        meq[8] = jx * jy;
        if (promote) {
            /* MRT_GPU.py:638-642: the literals -2.0, 3.0, 9.0 make these sums double; the products of jx, jy among themselves are float */
            const REAL s = jx * jx + jy * jy, p4 = ((jx * jx) * jy) * jy, x3 = (jx * jx) * jx, y3 = (jy * jy) * jy;
            meq[1] = (REAL)(-2.0 * (double)rho + 3.0 * (double)s);
            meq[2] = (REAL)((-3.0 * (double)s + (double)rho) + 9.0 * (double)p4);
            meq[4] = (REAL)((double)(-jx) + 3.0 * (double)x3);
            meq[6] = (REAL)((double)(-jy) + 3.0 * (double)y3);
        }
        for (int k = 0; k < 9; ++k) m[k] = m[k] - wv[k] * (m[k] - meq[k]);
        for (int k = 0; k < 9; ++k) {
            REAL acc = mi[k][0] * m[0];
            for (int j = 1; j < 9; ++j) acc = acc + mi[k][j] * m[j];
            out[k] = acc;
        }
    }
}

#ifdef LBMREF_DEFINE_THREADS
/* Threads used by the "omp parallel for" loops below (results do not depend on it:
 * every pass is elementwise or a gather between distinct arrays).  Default 1. */
static int g_threads = 1;
void lbmref_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int lbmref_get_threads(void) { return g_threads; }
int lbmref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}
#else
extern int lbmref_get_threads(void);
#endif

/* Advance nsteps iterations in place.  fin[9][nx][ny]; rho[nx][ny], u[2][nx][ny] receive the
 * macroscopic fields computed in the LAST iteration (MRT.py:500-503 one-step lag).
 * Returns 0, or -1 on allocation failure / bad arguments. */
int FN(lbmref_step)(REAL* fin, REAL* rho_out, REAL* u_out, int nx, int ny, int nsteps, int semantics,
                    int collision, const double* relax, double uLB_d, int turb, REAL* feq_hist, int promote) {
    /* turb != 0: feq_hist[9][nx][ny] holds the previous step's equilibrium on entry (initially a
     * copy of fin, MRT_GPU.py:325) and rho_out the previous step's density (initially 1); both are
     * updated in place. */
    if (nx < 4 || ny < 4 || nsteps < 0 || (turb && (!feq_hist || semantics != SEM_MRT_GPU))) return -1;
    if (promote && semantics != SEM_MRT_GPU) return -1;   /* MRT.py is NumPy fp64: nothing to promote */
    const size_t n = (size_t)nx * ny;
    REAL* fpost = (REAL*)malloc(9 * n * sizeof(REAL));
    REAL* feq = (REAL*)malloc(9 * n * sizeof(REAL));
    if (!fpost || !feq) { free(fpost); free(feq); return -1; }
    REAL t[9], w[5], mi[9][9];
    t[0] = (REAL)(4.0 / 9.0);
    for (int k = 1; k < 5; ++k) t[k] = (REAL)(1.0 / 9.0);
    for (int k = 5; k < 9; ++k) t[k] = (REAL)(1.0 / 36.);
    for (int i = 0; i < 5; ++i) w[i] = (REAL)relax[i];
    for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) mi[a][b] = (REAL)MINV[a][b];
    const REAL uLB = (REAL)uLB_d;
    const int X = nx, Y = ny;
#ifdef _OPENMP
    omp_set_num_threads(lbmref_get_threads());
#endif
#define F(a, k, x, y) (a)[(size_t)(k) * n + (size_t)(x) * Y + (y)]

    for (int it = 0; it < nsteps; ++it) {
        /* pass 1: a4 moments, a5 overrides, a3 equilibrium, a6 collide */
#pragma omp parallel for schedule(static)
        for (int x = 0; x < X; ++x) {
            for (int y = 0; y < Y; ++y) {
                REAL f[9], fe[9], fo[9];
                for (int k = 0; k < 9; ++k) f[k] = F(fin, k, x, y);
                REAL rho = ((((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7]) + f[8]);
                REAL ux = (((((f[1] - f[3]) + f[5]) - f[6]) - f[7]) + f[8]) / rho; /* MRT.py:320 */
                REAL uy = (((((f[2] - f[4]) + f[5]) + f[6]) - f[7]) - f[8]) / rho; /* MRT.py:321 */
                if (x == 0 || x == X - 1 || y == Y - 1) { ux = 0; uy = 0; }       /* MRT.py:341 */
                if (y == 0) {                                                      /* MRT.py:337,342 */
                    rho = ((f[0] + f[1]) + f[3]) + (REAL)2. * ((f[2] + f[5]) + f[6]);
                    ux = uLB; uy = 0;
                }
                REAL wc[5] = {w[0], w[1], w[2], w[3], w[4]};
                if (turb) {
                    REAL fp[9];
                    for (int k = 0; k < 9; ++k) fp[k] = F(feq_hist, k, x, y);
                    wc[0] = smagorinsky_omega(f, fp, rho_out[(size_t)x * Y + y], w[0], promote);
                }
                equ_cell(rho, ux, uy, t, fe, promote);
                collide_cell(collision, f, rho, fe, wc, mi, fo, promote);
                for (int k = 0; k < 9; ++k) { F(feq, k, x, y) = fe[k]; F(fpost, k, x, y) = fo[k]; }
                rho_out[(size_t)x * Y + y] = rho;
                u_out[(size_t)x * Y + y] = ux;
                u_out[n + (size_t)x * Y + y] = uy;
            }
        }
        /* pass 2: a7 streaming, destination windows */
        for (int k = 0; k < 9; ++k) {
            int cx = CX[k], cy = CY[k], x0, x1, y0, y1;
            if (semantics == SEM_MRT_PY) { /* MRT.py:404-414 */
                x0 = cx > 0 ? 1 : 0; x1 = cx > 0 ? X - 2 : (cx < 0 ? X - 3 : X - 1);
                y0 = cy < 0 ? 1 : 0; y1 = cy > 0 ? Y - 3 : (cy < 0 ? Y - 2 : Y - 1);
            } else { /* MRT_GPU.py:412 */
                x0 = cx > 0 ? 1 : 0; x1 = cx < 0 ? X - 2 : X - 1;
                y0 = cy < 0 ? 1 : 0; y1 = cy > 0 ? Y - 2 : Y - 1;
            }
#pragma omp parallel for schedule(static)
            for (int x = x0; x <= x1; ++x)
                for (int y = y0; y <= y1; ++y) F(fin, k, x, y) = F(fpost, k, x - cx, y + cy);
        }
        /* pass 3: a8 wall boundary conditions */
        if (semantics == SEM_MRT_PY) { /* MRT.py:450-453, statement order matters at corners */
            static const int R_[3] = {1, 5, 8}, L_[3] = {3, 6, 7}, T_[3] = {2, 5, 6}, B_[3] = {4, 7, 8};
            for (int y = 0; y < Y; ++y)
                for (int i = 0; i < 3; ++i) F(fin, R_[i], 0, y) = F(feq, R_[i], 0, y);
            for (int y = 0; y < Y; ++y) {
                REAL v[3];
                for (int i = 0; i < 3; ++i)
                    v[i] = -F(feq, R_[i], X - 1, y) + (F(feq, L_[i], X - 1, y) + F(fin, R_[i], X - 1, y));
                for (int i = 0; i < 3; ++i) F(fin, L_[i], X - 1, y) = v[i];
            }
            for (int x = 0; x < X; ++x) {
                REAL v[3];
                for (int i = 0; i < 3; ++i)
                    v[i] = -F(feq, B_[i], x, Y - 1) + (F(feq, T_[i], x, Y - 1) + F(fin, B_[i], x, Y - 1));
                for (int i = 0; i < 3; ++i) F(fin, T_[i], x, Y - 1) = v[i];
            }
            for (int x = 0; x < X; ++x) {
                REAL v[3];
                for (int i = 0; i < 3; ++i)
                    v[i] = -F(feq, T_[i], x, 0) + (F(feq, B_[i], x, 0) + F(fin, T_[i], x, 0));
                for (int i = 0; i < 3; ++i) F(fin, B_[i], x, 0) = v[i];
            }
        } else { /* MRT_GPU.py:674-692: per cell, x rule then y rule, sequential statements */
            for (int x = 0; x < X; ++x) {
                for (int y = 0; y < Y; ++y) {
                    if (x != 0 && x != X - 1 && y != 0 && y != Y - 1) continue;
                    if (x == 0) {
                        F(fin, 1, x, y) = (F(feq, 1, x, y) - F(feq, 3, x, y)) + F(fin, 3, x, y);
                        F(fin, 5, x, y) = (F(feq, 5, x, y) - F(feq, 7, x, y)) + F(fin, 7, x, y);
                        F(fin, 8, x, y) = (F(feq, 8, x, y) - F(feq, 6, x, y)) + F(fin, 6, x, y);
                    } else if (x == X - 1) {
                        F(fin, 3, x, y) = (-F(feq, 1, x, y) + F(feq, 3, x, y)) + F(fin, 1, x, y);
                        F(fin, 6, x, y) = (-F(feq, 8, x, y) + F(feq, 6, x, y)) + F(fin, 8, x, y);
                        F(fin, 7, x, y) = (-F(feq, 5, x, y) + F(feq, 7, x, y)) + F(fin, 5, x, y);
                    }
                    if (y == Y - 1) {
                        F(fin, 2, x, y) = (-F(feq, 4, x, y) + F(feq, 2, x, y)) + F(fin, 4, x, y);
                        F(fin, 5, x, y) = (-F(feq, 7, x, y) + F(feq, 5, x, y)) + F(fin, 7, x, y);
                        F(fin, 6, x, y) = (-F(feq, 8, x, y) + F(feq, 6, x, y)) + F(fin, 8, x, y);
                    } else if (y == 0) {
                        F(fin, 4, x, y) = (-F(feq, 2, x, y) + F(feq, 4, x, y)) + F(fin, 2, x, y);
                        F(fin, 7, x, y) = (-F(feq, 5, x, y) + F(feq, 7, x, y)) + F(fin, 5, x, y);
                        F(fin, 8, x, y) = (-F(feq, 6, x, y) + F(feq, 8, x, y)) + F(fin, 6, x, y);
                    }
                }
            }
        }
        if (turb) memcpy(feq_hist, feq, 9 * n * sizeof(REAL));
    }
#undef F
    free(fpost);
    free(feq);
    return 0;
}

/* equilibrium of the macroscopic state (with wall overrides) of given populations: the
 * Smagorinsky history after a state upload (same convention as lbm_set_state) */
int FN(lbmref_history)(const REAL* fin, REAL* rho_out, REAL* feq_hist, int nx, int ny, double uLB_d, int promote) {
    const size_t n = (size_t)nx * ny;
    REAL t[9];
    t[0] = (REAL)(4.0 / 9.0);
    for (int k = 1; k < 5; ++k) t[k] = (REAL)(1.0 / 9.0);
    for (int k = 5; k < 9; ++k) t[k] = (REAL)(1.0 / 36.);
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            REAL f[9], fe[9];
            for (int k = 0; k < 9; ++k) f[k] = fin[(size_t)k * n + (size_t)x * ny + y];
            REAL rho = ((((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7]) + f[8]);
            REAL ux = (((((f[1] - f[3]) + f[5]) - f[6]) - f[7]) + f[8]) / rho;
            REAL uy = (((((f[2] - f[4]) + f[5]) + f[6]) - f[7]) - f[8]) / rho;
            if (x == 0 || x == nx - 1 || y == ny - 1) { ux = 0; uy = 0; }
            if (y == 0) { rho = ((f[0] + f[1]) + f[3]) + (REAL)2. * ((f[2] + f[5]) + f[6]); ux = (REAL)uLB_d; uy = 0; }
            equ_cell(rho, ux, uy, t, fe, promote);
            for (int k = 0; k < 9; ++k) feq_hist[(size_t)k * n + (size_t)x * ny + y] = fe[k];
            rho_out[(size_t)x * ny + y] = rho;
        }
    return 0;
}

/* A.7 init: fin = equ(rho = 1, u = (uLB on the lid row, 0))  (MRT.py:206,260-268).  promote: MRT_GPU.py:230-247 evaluates the same
 * formula on the host with cu, usqr in float64 arrays and stores into float32 -- the bracket in double, one rounding -- which for
 * rho = 1 is the device formula above (with the NumPy of 2017, float32 array * float64 scalar = float32: rho*t[i] is a float). */
int FN(lbmref_init)(REAL* fin, int nx, int ny, double uLB_d, int promote) {
    const size_t n = (size_t)nx * ny;
    REAL t[9], fe[9];
    t[0] = (REAL)(4.0 / 9.0);
    for (int k = 1; k < 5; ++k) t[k] = (REAL)(1.0 / 9.0);
    for (int k = 5; k < 9; ++k) t[k] = (REAL)(1.0 / 36.);
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            equ_cell((REAL)1, y == 0 ? (REAL)uLB_d : (REAL)0, (REAL)0, t, fe, promote);
            for (int k = 0; k < 9; ++k) fin[(size_t)k * n + (size_t)x * ny + y] = fe[k];
        }
    return 0;
}
