// lbm_device.hpp -- device-side D2Q9 operators shared by every kernel of liblbm_hip.
//
// Arithmetic contract: compiled with -ffp-contract=off, every + - * / below is ONE IEEE
// operation in the order written, so results are bit-identical to a scalar CPU evaluation
// of the same expressions (the parity tests rely on this).  Operation order follows the
// reference scripts; each operator cites the lines it restates.
#pragma once
#include <hip/hip_runtime.h>

namespace lbm {

constexpr int Q = 9;
constexpr int GH = 4;  // ghost/pad columns on each side of a row (keeps x = 0 16-byte aligned)
constexpr int GHY = 10; // ghost rows above and below a lattice: row -1 / ny is the one-row halo of a slab (and holds parked wall data);
                        // rows -S .. -1 / ny .. ny+S-1 receive the neighbour's rows for the multi-step launches (deep halo, S <= 8)

// a1: lattice vectors, MRT.py:138-140 (k: 0 rest, 1 E, 2 N, 3 W, 4 S, 5 NE, 6 NW, 7 SW, 8 SE).
// A population moves from (x, y) to (x + cx, y - cy): y = 0 is the lid (MRT_GPU.py:412-413).
__host__ __device__ constexpr int cxk(int k) { return k == 1 || k == 5 || k == 8 ? 1 : (k == 3 || k == 6 || k == 7 ? -1 : 0); }
__host__ __device__ constexpr int cyk(int k) { return k == 2 || k == 5 || k == 6 ? 1 : (k == 4 || k == 7 || k == 8 ? -1 : 0); }

enum { SEM_PY = 0, SEM_GPU = 1 };
enum { C_SRT = 0, C_TRT = 1, C_MRT = 2,
       // lbm_params.arith = LBM_ARITH_FAST: not the reference's operation order / rounding --
       C_MRT_FAST = 3,     // the MRT operator in factored form with fused multiply-adds
       C_SRT_FAST = 4,     // SRT / TRT as in the strict form, but u = j * rcp(rho) and the closure's divisions and square root by
       C_TRT_FAST = 5 };   // v_rcp_f32 / v_sqrt_f32 (1 ulp; fp64: v_rcp_f64 / v_rsq_f64 + Newton) instead of the IEEE sequences
constexpr bool coll_is_mrt(int c) { return c == C_MRT || c == C_MRT_FAST; }
constexpr bool coll_is_fast(int c) { return c >= C_MRT_FAST; }

template <typename R>
struct Relax {  // a2: MRT_GPU.py:63-93
    R uLB, w_nu, w_m, w_e, w_eps, w_q;
};

// The operators below are written once for a value type T that is either a scalar real or a 2-wide extended vector of
// reals (two cells side by side): on gfx950 fp32 pairs compile to packed v_pk_add_f32 / v_pk_mul_f32, lane-wise IEEE, so the
// results are the same bits as the scalar evaluation.  S = ScalarOf<T>::type is the type of the constants.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct ScalarOf { typedef T type; };
template <> struct ScalarOf<f32x2> { typedef float type; };

// Destination window of direction k (a7).  SEM_PY: the truncated slices of MRT.py:404-414;
// SEM_GPU: "neighbour inside the lattice", MRT_GPU.py:412.  gy is the GLOBAL row.
template <int SEM>
__device__ __forceinline__ bool in_window(int k, int x, int gy, int X, int Y) {
    const int cx = cxk(k), cy = cyk(k);
    bool ok = true;
    if (SEM == SEM_PY) {
        if (cx > 0) ok = ok && (x >= 1) && (x <= X - 2);
        if (cx < 0) ok = ok && (x <= X - 3);
        if (cy > 0) ok = ok && (gy <= Y - 3);
        if (cy < 0) ok = ok && (gy >= 1) && (gy <= Y - 2);
    } else {
        if (cx > 0) ok = ok && (x >= 1);
        if (cx < 0) ok = ok && (x <= X - 2);
        if (cy > 0) ok = ok && (gy <= Y - 2);
        if (cy < 0) ok = ok && (gy >= 1);
    }
    return ok;
}

// a3: equilibrium, MRT.py:213-231 / MRT_GPU.py:408-410.  cu is formed without the
// multiplications by 0 and +-1 of the reference (value-identical in IEEE arithmetic).
template <typename T>
__device__ __forceinline__ T cu_of(int k, T ux, T uy) {
    switch (k) {
        case 1: return ux;
        case 2: return uy;
        case 3: return -ux;
        case 4: return -uy;
        case 5: return ux + uy;
        case 6: return -ux + uy;
        case 7: return -ux + -uy;
        case 8: return ux + -uy;
        default: return T{};
    }
}
template <typename R>
__device__ __forceinline__ R weight(int k) {
    return k == 0 ? (R)(4.0 / 9.0) : (k < 5 ? (R)(1.0 / 9.0) : (R)(1.0 / 36.));
}
// a / b and sqrt: IEEE-exact, or (FAST, fp32 only) by v_rcp_f32 / v_sqrt_f32 (1 ulp) -- the same instruction per lane in the
// scalar and the packed form, so every kernel variant still produces the same bits
template <bool FAST> __device__ __forceinline__ float div_(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
// fp64: v_rcp_f64 / v_rsq_f64 (about 2^-26) refined by two Newton steps (error far below one ulp of the iterate; the product with
// `a` rounds once more) instead of the IEEE sequences of ~25 instructions
__device__ __forceinline__ double rcp_nr(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return r;
}
template <bool FAST> __device__ __forceinline__ double div_(double a, double b) { return FAST ? a * rcp_nr(b) : a / b; }
template <bool FAST> __device__ __forceinline__ f32x2 div_(f32x2 a, f32x2 b) {
    return FAST ? a * f32x2{__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)} : a / b;
}
template <bool FAST> __device__ __forceinline__ float sqrt_(float a) { return FAST ? __builtin_amdgcn_sqrtf(a) : sqrtf(a); }
template <bool FAST> __device__ __forceinline__ double sqrt_(double a) {
    if (!FAST) return sqrt(a);
    if (a <= 0.0) return 0.0;                        // (the closure's argument is tau0^2 + ... > 0)
    double y = __builtin_amdgcn_rsq(a);              // 1 / sqrt(a), ~2^-26
    y = y * __builtin_fma(-0.5 * a * y, y, 1.5);     // Newton on 1 / sqrt
    y = y * __builtin_fma(-0.5 * a * y, y, 1.5);
    const double g = a * y;                          // sqrt(a), then one correction step
    return __builtin_fma(__builtin_fma(-g, g, a), 0.5 * y, g);
}
template <bool FAST> __device__ __forceinline__ f32x2 sqrt_(f32x2 a) {
    return FAST ? f32x2{__builtin_amdgcn_sqrtf(a.x), __builtin_amdgcn_sqrtf(a.y)} : f32x2{sqrtf(a.x), sqrtf(a.y)};
}

template <typename T>
__device__ __forceinline__ void equ(T rho, T ux, T uy, T (&feq)[Q]) {
    typedef typename ScalarOf<T>::type S;
    const T usqr = ux * ux + uy * uy;
    const T c = (S)1.5 * usqr;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const T cu = cu_of<T>(k, ux, uy);
        feq[k] = (rho * weight<S>(k)) * ((((S)1. + (S)3.0 * cu) + ((S)4.5 * cu) * cu) - c);
    }
}

// a4 + a5: moments with the macroscopic wall overrides (MRT.py:292,320-321,337,341-342;
// MRT_GPU.py:389-405).  rho_sum is the plain sum (used by nothing but kept for clarity).
template <typename R, bool FAST = false>
__device__ __forceinline__ void macros(const R (&f)[Q], int x, int gy, int X, int Y, R uLB, R& rho, R& ux, R& uy) {
    rho = ((((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7]) + f[8]);
    ux = div_<FAST>((((((f[1] - f[3]) + f[5]) - f[6]) - f[7]) + f[8]), rho);
    uy = div_<FAST>((((((f[2] - f[4]) + f[5]) + f[6]) - f[7]) - f[8]), rho);
    if (x == 0 || x == X - 1 || gy == Y - 1) { ux = (R)0; uy = (R)0; }
    if (gy == 0) {
        rho = ((f[0] + f[1]) + f[3]) + (R)2. * ((f[2] + f[5]) + f[6]);
        ux = uLB; uy = (R)0;
    }
}

// fused multiply-add on a scalar real or on two packed cells (v_fma_f32 / v_fma_f64 / v_pk_fma_f32): one rounding, lane-wise
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ f32x2 fma_(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// a6: collision.  SRT MRT.py:396 / MRT_GPU.py:413; TRT MRT_GPU.py:455-462,514-525;
// MRT MRT_GPU.py:633-655 (m = M f with jx = m3, jy = m5 from the raw populations, the
// reference's own m_eq polynomial, f* = Minv m).  Zero matrix entries are skipped and the
// +-1, +-2, +-4 entries of M are exact scalings: value-identical to the dense products.
template <typename T, int COLL>
__device__ __forceinline__ void collide(const T (&f)[Q], T rho, const T (&feq)[Q],
                                        const Relax<typename ScalarOf<T>::type>& w, T w_nu, T (&out)[Q]) {
    // w_nu: the viscous rate of each cell (w.w_nu, or the Smagorinsky value); the other rates are lattice-wide scalars
    typedef typename ScalarOf<T>::type R;
    if (COLL == C_SRT || COLL == C_SRT_FAST) {
#pragma unroll
        for (int k = 0; k < Q; ++k) out[k] = f[k] - w_nu * (f[k] - feq[k]);
    } else if (COLL == C_TRT || COLL == C_TRT_FAST) {
        T fp[Q], fm[Q], ep[Q], em[Q];
        constexpr int pa[4] = {2, 5, 6, 1}, pb[4] = {4, 7, 8, 3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a = pa[i], b = pb[i];
            fp[a] = (R)0.5 * (f[a] + f[b]); fp[b] = fp[a];
            fm[a] = (R)0.5 * (f[a] - f[b]); fm[b] = -fm[a];
            ep[a] = (R)0.5 * (feq[a] + feq[b]); ep[b] = ep[a];
            em[a] = (R)0.5 * (feq[a] - feq[b]); em[b] = -em[a];
        }
        fp[0] = f[0]; fm[0] = T{}; ep[0] = feq[0]; em[0] = T{};
#pragma unroll
        for (int k = 0; k < Q; ++k) out[k] = (f[k] - w_nu * (fp[k] - ep[k])) - w.w_m * (fm[k] - em[k]);
    } else if (COLL == C_MRT_FAST) {
        // The same operator, m = M f, m* = m - S (m - m_eq), f* = Minv m*, with the sums of M and Minv factored through the
        // pairs f1 +- f3, f2 +- f4, f5 +- f7, f6 +- f8 and explicit fused multiply-adds: ~75 operations per cell instead of
        // ~190.  Algebraically identical to the branch below, NOT the reference's operation order: results differ from the
        // strict form in the last bits (tolerances in tests/test_gpu_parity.py::test_fast_arithmetic_*).  The operations are
        // spelled out (no compiler contraction), so every kernel variant performs the same ones and a lattice gives the same
        // bits however it is cut into tiles, frames or slabs.
        const T a13 = f[1] + f[3], d13 = f[1] - f[3], a24 = f[2] + f[4], d24 = f[2] - f[4];
        const T a57 = f[5] + f[7], d57 = f[5] - f[7], a68 = f[6] + f[8], d68 = f[6] - f[8];
        const T sa = a13 + a24, sd = a57 + a68, dm = d57 - d68, dp = d57 + d68;
        const T r = (f[0] + sa) + sd;                                       // m0
        const T jx = d13 + dm, jy = d24 + dp;                               // m3, m5
        const T f04 = (R)4 * f[0];
        T e = fma_(T((R)2), sd, -sa) - f04;                                 // m1
        T eps = fma_(T((R)-2), sa, f04 + sd);                               // m2
        T qx = fma_(T((R)-2), d13, dm), qy = fma_(T((R)-2), d24, dp);       // m4, m6
        T pxx = a13 - a24, pxy = a57 - a68;                                 // m7, m8
        const T jx2 = jx * jx, jy2 = jy * jy, j23 = (R)3 * (jx2 + jy2);
        e = fma_(T(-w.w_e), e - fma_(T((R)-2), r, j23), e);
        eps = fma_(T(-w.w_eps), eps - fma_(T((R)9), jx2 * jy2, r - j23), eps);
        qx = fma_(T(-w.w_q), qx - jx * fma_(T((R)3), jx2, T((R)-1)), qx);
        qy = fma_(T(-w.w_q), qy - jy * fma_(T((R)3), jy2, T((R)-1)), qy);
        pxx = fma_(-w_nu, pxx - (jx2 - jy2), pxx);
        pxy = fma_(-w_nu, fma_(-jx, jy, pxy), pxy);
        const R a9 = (R)(1.0 / 9), a36 = (R)(1.0 / 36), a18 = (R)(1.0 / 18), a6 = (R)(1.0 / 6),
                a12 = (R)(1.0 / 12), a4 = (R)(1.0 / 4);
        const T r9 = a9 * r;
        out[0] = a9 * ((r - e) + eps);
        const T A = fma_(T(-a18), eps, fma_(T(-a36), e, r9)), D = fma_(T(a36), eps, fma_(T(a18), e, r9));
        const T Ap = fma_(T(a4), pxx, A), Am = fma_(T(-a4), pxx, A), Dp = fma_(T(a4), pxy, D), Dm = fma_(T(-a4), pxy, D);
        const T bx = a6 * (jx - qx), by = a6 * (jy - qy);
        const T X = fma_(T(a12), qx, a6 * jx), Y = fma_(T(a12), qy, a6 * jy);
        const T XpY = X + Y, XmY = X - Y;
        out[1] = Ap + bx; out[3] = Ap - bx;
        out[2] = Am + by; out[4] = Am - by;
        out[5] = Dp + XpY; out[7] = Dp - XpY;
        out[8] = Dm + XmY; out[6] = Dm - XmY;
    } else {
        T m[Q], meq[Q];
        // rows of M_GS (MRT.py:163-173), left-to-right sums.  A term  acc + c * x  with c = +-2, +-4 or +-1/4 is written as ONE fused
        // multiply-add: the product by a power of two is exact, so round(acc + c x) is the same number whether the product is
        // rounded first (the reference: multiply, then add) or not -- same bits, one instruction instead of two (19 of ~160
        // operations per cell; 4096^2 strict: fp32 247 -> 270 GLUPS, fp64 118 -> 125, every test still bit-identical to the oracle).  Products by 3, 9, 1/9 ... are not exact and stay two operations.
        const T c2 = T((R)2), cm2 = T((R)-2);
        m[0] = (((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7]) + f[8];
        m[1] = fma_(c2, f[8], fma_(c2, f[7], fma_(c2, f[6], fma_(c2, f[5], ((((R)-4 * f[0] - f[1]) - f[2]) - f[3]) - f[4]))));
        m[2] = (((fma_(cm2, f[4], fma_(cm2, f[3], fma_(cm2, f[2], fma_(cm2, f[1], (R)4 * f[0])))) + f[5]) + f[6]) + f[7]) + f[8];
        m[3] = ((((f[1] - f[3]) + f[5]) - f[6]) - f[7]) + f[8];
        m[4] = (((fma_(c2, f[3], (R)-2 * f[1]) + f[5]) - f[6]) - f[7]) + f[8];
        m[5] = ((((f[2] - f[4]) + f[5]) + f[6]) - f[7]) - f[8];
        m[6] = (((fma_(c2, f[4], (R)-2 * f[2]) + f[5]) + f[6]) - f[7]) - f[8];
        m[7] = ((f[1] - f[2]) + f[3]) - f[4];
        m[8] = ((f[5] - f[6]) + f[7]) - f[8];
        const T jx = m[3], jy = m[5];
        meq[0] = rho;
        meq[1] = fma_(cm2, rho, (R)3.0 * (jx * jx + jy * jy));
        meq[2] = ((R)-3.0 * (jx * jx + jy * jy) + rho) + (R)9.0 * (((jx * jx) * jy) * jy);
        meq[3] = m[3];
        meq[4] = -jx + (R)3.0 * ((jx * jx) * jx);
        meq[5] = m[5];
        meq[6] = -jy + (R)3.0 * ((jy * jy) * jy);
        meq[7] = jx * jx - jy * jy;
        meq[8] = jx * jy;
        // The reference relaxes all nine moments, three of them with rate 0 (MRT.py:141, MRT_GPU.py:650: m[k] - 0 * (m[k] - meq[k]) for the
        // density and the two momenta).  That is m[k] itself, bit for bit, for every finite input: 0 * x = +-0 and m - (+-0) = m unless m is
        // a zero of the other sign -- and there meq[k] = m[k] (k = 3, 5: x - x = +0, so the product is +0 and -0 - (+0) = -0 stays), while
        // m[0] is a positive density.  Left out: nine operations of ~150 per cell (4096^2 strict fp32 302 -> 307 GLUPS, fp64 136 -> 141), every strict test still
        // np.array_equal to the oracle, which spells the three out.
        m[1] = m[1] - w.w_e * (m[1] - meq[1]);
        m[2] = m[2] - w.w_eps * (m[2] - meq[2]);
        m[4] = m[4] - w.w_q * (m[4] - meq[4]);
        m[6] = m[6] - w.w_q * (m[6] - meq[6]);
        m[7] = m[7] - w_nu * (m[7] - meq[7]);
        m[8] = m[8] - w_nu * (m[8] - meq[8]);
        // rows of M_GS_INV (MRT.py:175-183)
        const R a9 = (R)(1.0 / 9), a36 = (R)(1.0 / 36), a18 = (R)(1.0 / 18), a6 = (R)(1.0 / 6),
                a12 = (R)(1.0 / 12), a4 = (R)(1.0 / 4);
        const T q4 = T(a4), qm4 = T(-a4);
        out[0] = (a9 * m[0] + -a9 * m[1]) + a9 * m[2];
        out[1] = fma_(q4, m[7], (((a9 * m[0] + -a36 * m[1]) + -a18 * m[2]) + a6 * m[3]) + -a6 * m[4]);
        out[2] = fma_(qm4, m[7], (((a9 * m[0] + -a36 * m[1]) + -a18 * m[2]) + a6 * m[5]) + -a6 * m[6]);
        out[3] = fma_(q4, m[7], (((a9 * m[0] + -a36 * m[1]) + -a18 * m[2]) + -a6 * m[3]) + a6 * m[4]);
        out[4] = fma_(qm4, m[7], (((a9 * m[0] + -a36 * m[1]) + -a18 * m[2]) + -a6 * m[5]) + a6 * m[6]);
        out[5] = fma_(q4, m[8], (((((a9 * m[0] + a18 * m[1]) + a36 * m[2]) + a6 * m[3]) + a12 * m[4]) + a6 * m[5]) + a12 * m[6]);
        out[6] = fma_(qm4, m[8], (((((a9 * m[0] + a18 * m[1]) + a36 * m[2]) + -a6 * m[3]) + -a12 * m[4]) + a6 * m[5]) + a12 * m[6]);
        out[7] = fma_(q4, m[8], (((((a9 * m[0] + a18 * m[1]) + a36 * m[2]) + -a6 * m[3]) + -a12 * m[4]) + -a6 * m[5]) + -a12 * m[6]);
        out[8] = fma_(qm4, m[8], (((((a9 * m[0] + a18 * m[1]) + a36 * m[2]) + a6 * m[3]) + a12 * m[4]) + -a6 * m[5]) + -a12 * m[6]);
    }
}

// Smagorinsky sub-grid closure, MRT_GPU.py:368-387 (turb = 1).  The Van Driest lines 370-373 are
// dead code (Cs2 is overwritten by 0.025 at line 374).  The reference reads, per cell, the
// equilibrium and the density that funRT wrote in the PREVIOUS step; only sum_k cx cy feq_k of it
// is used, so the history kept per cell is two reals: planes Q (that sum) and Q + 1 (density).
constexpr int K_QEQ = Q;      // plane index of  -feq8 + (feq7 + (-feq6 + feq5))  of the previous step
constexpr int K_RHO = Q + 1;  // plane index of the previous step's density (after the lid override)

template <typename T>
__device__ __forceinline__ T diag_flux(const T (&f)[Q]) {   // product = cx cy f_k + product, k = 0..8
    return -f[8] + (f[7] + (-f[6] + f[5]));
}
// Equilibrium + collision of a cell (or of two packed cells) from its macroscopic state, as every step kernel calls it; q2 (TURB only):
// sum_k cx cy feq_k of THIS step, the closure's history (K_QEQ).  The strict operators and the factored MRT one: equ() -> collide() ->
// diag_flux(), exactly the sequence the kernels spelled out before.  C_SRT_FAST / C_TRT_FAST (r03): the same operators with the
// equilibrium never formed -- feq_k = rho t_k (B + 4.5 cu^2 +- 3 cu), B = 1 - 1.5 u^2, shares everything but the sign of the odd part
// between the two directions of a pair, so
//   SRT  f*_k = (1 - w) f_k + (w rho t_k) (E_k +- O_k),                      E = B + 4.5 cu^2, O = 3 cu
//   TRT  f*_a/b = f_a/b - P -+ M,  P = w+ ((f_a + f_b) / 2 - rho t E),  M = w- ((f_a - f_b) / 2 - rho t O)
// with fused multiply-adds: ~50 / ~65 operations per cell after the moments instead of ~105 / ~140, and sum_k cx cy feq_k = rho ux uy
// exactly.  Same operator, other rounding: NOT the reference's operation order (tolerances: tests/test_gpu_parity.py::test_fast_arithmetic_*);
// one spelling for every kernel, so a lattice gives the same bits however it is cut.
template <typename T, int COLL, bool TURB>
__device__ __forceinline__ void equ_collide(const T (&g)[Q], T rho, T ux, T uy, const Relax<typename ScalarOf<T>::type>& w, T w_nu, T (&out)[Q], T& q2) {
    typedef typename ScalarOf<T>::type R;
    // (fp32 SRT with the closure keeps the unfused sequence: at the streaming kernels' 128 registers the fused one spills there -- 4096^2
    // 256 -> 210 GLUPS, while fp64 gains, 108 -> 134; without the closure it is what lets fp32 SRT take the walls kernel, 251 -> 414)
    if constexpr ((COLL == C_SRT_FAST && !(TURB && sizeof(R) == 4)) || COLL == C_TRT_FAST) {
        const T base = fma_(T((R)-1.5), fma_(uy, uy, ux * ux), T((R)1));
        const T orho = w_nu * rho, c1 = (R)1 - w_nu;
        const T r1 = (R)(1.0 / 9) * orho, r5 = (R)(1.0 / 36) * orho;
        out[0] = fma_((R)(4.0 / 9) * orho, base, c1 * g[0]);
        if constexpr (COLL == C_SRT_FAST) {
            auto pair = [&](T cu, T rw, const T& ga, const T& gb, T& oa, T& ob) {
                const T e = fma_((R)4.5 * cu, cu, base), o = (R)3 * cu;
                oa = fma_(rw, e + o, c1 * ga);
                ob = fma_(rw, e - o, c1 * gb);
            };
            pair(ux, r1, g[1], g[3], out[1], out[3]);
            pair(uy, r1, g[2], g[4], out[2], out[4]);
            pair(ux + uy, r5, g[5], g[7], out[5], out[7]);
            pair(ux - uy, r5, g[8], g[6], out[8], out[6]);
        } else {
            const T mrho = w.w_m * rho, hp = (R)0.5 * w_nu;
            const T m1 = (R)(1.0 / 9) * mrho, m5 = (R)(1.0 / 36) * mrho;
            const R hm = (R)0.5 * w.w_m;
            auto pair = [&](T cu, T rp, T rm, const T& ga, const T& gb, T& oa, T& ob) {
                const T e = fma_((R)4.5 * cu, cu, base), o = (R)3 * cu;
                const T P = fma_(-rp, e, hp * (ga + gb)), M = fma_(-rm, o, hm * (ga - gb));
                oa = (ga - P) - M;
                ob = (gb - P) + M;
            };
            pair(ux, r1, m1, g[1], g[3], out[1], out[3]);
            pair(uy, r1, m1, g[2], g[4], out[2], out[4]);
            pair(ux + uy, r5, m5, g[5], g[7], out[5], out[7]);
            pair(ux - uy, r5, m5, g[8], g[6], out[8], out[6]);
        }
        if (TURB) q2 = rho * (ux * uy);
    } else if constexpr (COLL == C_MRT_FAST) {
        // (the MRT operator needs neither u nor feq, MRT_GPU.py:633-648; the closure's history sum_k cx cy feq_k is rho ux uy, see above --
        // the nine equilibria the strict form builds for it are left out: 4096^2 MRT + closure fp32 212 -> 283 GLUPS, fp64 91 -> 117)
        T fe[Q];
        collide<T, COLL>(g, rho, fe, w, w_nu, out);
        if (TURB) q2 = rho * (ux * uy);
    } else {
        T fe[Q];
        if (!coll_is_mrt(COLL) || TURB) equ<T>(rho, ux, uy, fe);     // (the plain MRT operator needs neither u nor feq: MRT_GPU.py:633-648)
        collide<T, COLL>(g, rho, fe, w, w_nu, out);
        if (TURB) q2 = diag_flux<T>(fe);
    }
}

__device__ __forceinline__ float real_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double real_abs(double x) { return fabs(x); }
__device__ __forceinline__ f32x2 real_abs(f32x2 x) { return f32x2{fabsf(x.x), fabsf(x.y)}; }

// T: scalar real or f32x2; the result is the per-cell relaxation rate (a T, not a scalar)
template <typename T, bool FAST = false>
__device__ __forceinline__ T smagorinsky_tau(const T (&f)[Q], T qeq_prev, T rho_prev, typename ScalarOf<T>::type omega) {   // taus_g, MRT_GPU.py:385-387
    typedef typename ScalarOf<T>::type R;
    const R tau0 = (R)1.0 / omega;
    const T q = diag_flux<T>(f) - qeq_prev;
    return (R)0.5 * (tau0 + sqrt_<FAST>(tau0 * tau0 + div_<FAST>(((R)(18 * 1.4142) * (R)0.025) * real_abs(q), rho_prev)));
}
template <typename T, bool FAST = false>
__device__ __forceinline__ T smagorinsky_omega(const T (&f)[Q], T qeq_prev, T rho_prev, typename ScalarOf<T>::type omega) {
    typedef typename ScalarOf<T>::type R;
    return div_<FAST>(T((R)1.0), smagorinsky_tau<T, FAST>(f, qeq_prev, rho_prev, omega));
}

// a8: wall rules on the populations of ONE perimeter cell, given the equilibrium of the
// previous macroscopic state of that cell.  SEM_PY: MRT.py:450-453 in statement order
// (left equilibrium, right mirror pairing, bottom, lid); SEM_GPU: MRT_GPU.py:674-692
// (x rule if / else-if, then y rule if / else-if, opposite pairing).
template <typename R, int SEM>
__device__ __forceinline__ void wall_rules(R (&g)[Q], const R (&fe)[Q], int x, int gy, int X, int Y) {
    if (SEM == SEM_PY) {
        if (x == 0) { g[1] = fe[1]; g[5] = fe[5]; g[8] = fe[8]; }
        if (x == X - 1) {
            const R a = -fe[1] + (fe[3] + g[1]), b = -fe[5] + (fe[6] + g[5]), c = -fe[8] + (fe[7] + g[8]);
            g[3] = a; g[6] = b; g[7] = c;
        }
        if (gy == Y - 1) {
            const R a = -fe[4] + (fe[2] + g[4]), b = -fe[7] + (fe[5] + g[7]), c = -fe[8] + (fe[6] + g[8]);
            g[2] = a; g[5] = b; g[6] = c;
        }
        if (gy == 0) {
            const R a = -fe[2] + (fe[4] + g[2]), b = -fe[5] + (fe[7] + g[5]), c = -fe[6] + (fe[8] + g[6]);
            g[4] = a; g[7] = b; g[8] = c;
        }
    } else {
        if (x == 0) {
            g[1] = (fe[1] - fe[3]) + g[3];
            g[5] = (fe[5] - fe[7]) + g[7];
            g[8] = (fe[8] - fe[6]) + g[6];
        } else if (x == X - 1) {
            g[3] = (-fe[1] + fe[3]) + g[1];
            g[6] = (-fe[8] + fe[6]) + g[8];
            g[7] = (-fe[5] + fe[7]) + g[5];
        }
        if (gy == Y - 1) {
            g[2] = (-fe[4] + fe[2]) + g[4];
            g[5] = (-fe[7] + fe[5]) + g[7];
            g[6] = (-fe[8] + fe[6]) + g[8];
        } else if (gy == 0) {
            g[4] = (-fe[2] + fe[4]) + g[2];
            g[7] = (-fe[5] + fe[7]) + g[5];
            g[8] = (-fe[6] + fe[8]) + g[6];
        }
    }
}

// Addressing of the nine direction arrays.  Local row y in [-GHY, ny+GHY), column x in [-GH, nx+GH).
// Element (k, x, y) lives at k * plane + at(x, y).  Two layouts share this formula:
//   planes  [k][y][x]: plane = pitch * (ny + 2 GHY), row = pitch   (nine separate arrays)
//   rows    [y][k][x]: plane = pitch,            row = 9 * pitch   (the nine rows y of the nine
//                                                                   directions are adjacent)
// Every row of every direction is contiguous and 16-byte aligned in both.
struct Geo {
    long long plane;  // elements between the same cell of consecutive directions
    long long row;    // elements between consecutive rows of one direction
    int pitch;        // elements per row (nx + 2*GH, multiple of 4)
    int nx, ny;       // columns, local rows
    int y0, NY;       // first global row of this slab, global height
    __host__ __device__ __forceinline__ long long at(int x, int y) const { return (long long)(y + GHY) * row + GH + x; }
};

// The same (k, x, y) addressing for a small window of the lattice held in LDS (the fused frame passes keep the output of the
// intermediate passes there): window origin (x0, y0), `pitch` elements per row, `plane` per direction.  gather / update_cell
// take one addressing object for the source and one for the destination: Geo (a lattice in memory) or Window.
struct Window {
    int plane, pitch, x0, y0;
    __device__ __forceinline__ int at(int x, int y) const { return (y - y0) * pitch + (x - x0); }
};

// Where a perimeter cell parks the density of its last macroscopic state: slot 0 of the
// adjacent ghost cell (slot 0 of a ghost cell is never pulled).
template <typename A>
__device__ __forceinline__ auto wall_rho_at(const A& a, const Geo& g, int x, int y, int gy) -> decltype(a.at(x, y)) {
    if (gy == 0) return a.at(x, y - 1);
    if (gy == g.NY - 1) return a.at(x, y + 1);
    if (x == 0) return a.at(-1, y);
    return a.at(g.nx, y);
}

// Gather the post-stream, post-wall-rule populations of cell (x, y) from a lattice that
// holds post-collision values (+ kept slots + parked wall densities); raw != 0: the lattice
// holds plain populations (state just set by the host), nothing to stream.
template <typename R, int SEM, typename AS>
__device__ __forceinline__ void gather_a(const R* __restrict__ src, const AS& as, const Geo& geo, int raw, R uLB, int x, int y, R (&g)[Q]) {
    const int gy = geo.y0 + y;
    if (raw) {
#pragma unroll
        for (int k = 0; k < Q; ++k) g[k] = src[k * as.plane + as.at(x, y)];
        return;
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) g[k] = src[k * as.plane + as.at(x - cxk(k), y + cyk(k))];
    if (x == 0 || x == geo.nx - 1 || gy == 0 || gy == geo.NY - 1) {
        const R rho_w = src[wall_rho_at(as, geo, x, y, gy)];
        R fe[Q];
        equ<R>(rho_w, gy == 0 ? uLB : (R)0, (R)0, fe);
        wall_rules<R, SEM>(g, fe, x, gy, geo.nx, geo.NY);
    }
}
template <typename R, int SEM>
__device__ __forceinline__ void gather(const R* __restrict__ src, const Geo& geo, int raw, R uLB, int x, int y, R (&g)[Q]) {
    gather_a<R, SEM, Geo>(src, geo, geo, raw, uLB, x, y, g);
}

// One fused update of cell (x, y): gather -> (kept slots) -> moments -> collide -> store.
// as / ad: addressing of the source / destination (Geo = lattice in memory, Window = LDS window of the fused frame passes).
template <typename R, int COLL, int SEM, bool TURB, typename AS, typename AD>
__device__ __forceinline__ void update_cell_a(const R* __restrict__ src, const AS& as, R* __restrict__ dst, const AD& ad, const Geo& geo,
                                              const Relax<R>& w0, int raw, int x, int y) {
    const int X = geo.nx, Y = geo.NY, gy = geo.y0 + y;
    R g[Q];
    gather_a<R, SEM, AS>(src, as, geo, raw, w0.uLB, x, y, g);
    // kept slots: a slot outside its streaming window keeps its value; park it where the
    // next pull of this cell will look for it.
    const bool near_edge = (x <= 0) || (x >= X - 2) || (gy <= 0) || (gy >= Y - 2);
    if (near_edge) {
#pragma unroll
        for (int k = 1; k < Q; ++k)
            if (!in_window<SEM>(k, x, gy, X, Y)) dst[k * ad.plane + ad.at(x - cxk(k), y + cyk(k))] = g[k];
    }
    R rho, ux, uy, out[Q], q2 = (R)0;
    const auto me = ad.at(x, y);
    const auto me_s = as.at(x, y);
    const Relax<R>& w = w0;
    R w_nu = w0.w_nu;
    if (TURB) w_nu = smagorinsky_omega<R, coll_is_fast(COLL)>(g, src[K_QEQ * as.plane + me_s], src[K_RHO * as.plane + me_s], w0.w_nu);
    macros<R, coll_is_fast(COLL)>(g, x, gy, X, Y, w.uLB, rho, ux, uy);
    equ_collide<R, COLL, TURB>(g, rho, ux, uy, w, w_nu, out, q2);
    if (TURB) {
        dst[K_QEQ * ad.plane + me] = q2;
        dst[K_RHO * ad.plane + me] = rho;
    }
    if (!near_edge) {
#pragma unroll
        for (int k = 0; k < Q; ++k) dst[k * ad.plane + me] = out[k];
    } else {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            // skip the slot if the receiving cell exists but does not stream it (it is that
            // cell's kept slot, written by the receiving cell itself)
            const int dx = x + cxk(k), dgy = gy - cyk(k);
            const bool inside = dx >= 0 && dx < X && dgy >= 0 && dgy < Y;
            if (!inside || in_window<SEM>(k, dx, dgy, X, Y)) dst[k * ad.plane + me] = out[k];
        }
        if (x == 0 || x == X - 1 || gy == 0 || gy == Y - 1) dst[wall_rho_at(ad, geo, x, y, gy)] = rho;
    }
}

template <typename R, int COLL, int SEM, bool TURB = false>
__device__ __forceinline__ void update_cell(const R* __restrict__ src, R* __restrict__ dst, const Geo& geo,
                                            const Relax<R>& w0, int raw, int x, int y) {
    update_cell_a<R, COLL, SEM, TURB, Geo, Geo>(src, geo, dst, geo, geo, w0, raw, x, y);
}

// ------------------------------------------------------------------------------------------
// Vector path (MRT_GPU.py semantics only): a thread owns V consecutive cells of one row and
// moves 16 bytes per direction plane per access.  Rows on the lid / bottom wall are handed
// to update_cell (wall rules, kept slots, parked densities); on every other row the only
// wall cells are x = 0 and x = X-1, where MRT_GPU.py:674-682 reduces exactly to copying
// the opposite population: fe_a - fe_b is +0 for a resting wall (same weight, u = 0), so
// f_a = 0 + f_b.  Arithmetic per cell is the same sequence of IEEE operations as update_cell.
// ------------------------------------------------------------------------------------------
template <typename R, int V>
struct VecT {
    typedef R type __attribute__((ext_vector_type(V)));
};

template <typename R, int V, bool NT>
__device__ __forceinline__ typename VecT<R, V>::type vload(const R* p, bool aligned) {
    typedef typename VecT<R, V>::type T;
    if (aligned) {
        if (NT) return __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
        return *reinterpret_cast<const T*>(p);
    }
    T t;   // element-aligned only (x -+ 1 neighbours): one unaligned 16-byte global load
    __builtin_memcpy(&t, p, sizeof(T));
    return t;
}

template <typename R, int V, bool NT>
__device__ __forceinline__ void vstore(R* p, typename VecT<R, V>::type v) {
    typedef typename VecT<R, V>::type T;
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<T*>(p));
    else *reinterpret_cast<T*>(p) = v;
}

template <typename R, int COLL, int V, bool NT, bool TURB = false>
__device__ __forceinline__ void update_vec(const R* __restrict__ src, R* __restrict__ dst, const Geo& geo,
                                           const Relax<R>& w0, int raw, int x0, int y) {
    typedef typename VecT<R, V>::type T;
    const int X = geo.nx;
    const long long me = geo.at(x0, y);
    T in[Q], outv[Q], hq, hr;
    if (TURB) {
        hq = vload<R, V, NT>(src + K_QEQ * geo.plane + me, true);
        hr = vload<R, V, NT>(src + K_RHO * geo.plane + me, true);
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int cx = raw ? 0 : cxk(k), cy = raw ? 0 : cyk(k);
        in[k] = vload<R, V, NT>(src + k * geo.plane + geo.at(x0 - cx, y + cy), cxk(k) == 0 || raw);
    }
#pragma unroll
    for (int c = 0; c < V; ++c) {
        const int x = x0 + c;
        R g[Q], out[Q], q2 = (R)0;
#pragma unroll
        for (int k = 0; k < Q; ++k) g[k] = in[k][c];
        const bool left = (c == 0) && (x == 0), right = (c == V - 1) && (x == X - 1);
        if (!raw) {
            if (left) { g[1] = (R)0 + g[3]; g[5] = (R)0 + g[7]; g[8] = (R)0 + g[6]; }
            if (right) { g[3] = (R)0 + g[1]; g[6] = (R)0 + g[8]; g[7] = (R)0 + g[5]; }
        }
        R w_nu = w0.w_nu;
        if (TURB) w_nu = smagorinsky_omega<R, coll_is_fast(COLL)>(g, hq[c], hr[c], w0.w_nu);
        R rho = ((((((((g[0] + g[1]) + g[2]) + g[3]) + g[4]) + g[5]) + g[6]) + g[7]) + g[8]);
        R ux = (R)0, uy = (R)0;
        if (!coll_is_mrt(COLL) || TURB) {   // the plain MRT operator needs neither u nor feq (MRT_GPU.py:633-648)
            ux = div_<coll_is_fast(COLL)>((((((g[1] - g[3]) + g[5]) - g[6]) - g[7]) + g[8]), rho);
            uy = div_<coll_is_fast(COLL)>((((((g[2] - g[4]) + g[5]) + g[6]) - g[7]) - g[8]), rho);
            if (left || right) { ux = (R)0; uy = (R)0; }
        }
        equ_collide<R, COLL, TURB>(g, rho, ux, uy, w0, w_nu, out, q2);
#pragma unroll
        for (int k = 0; k < Q; ++k) outv[k][c] = out[k];
        if (TURB) { hq[c] = q2; hr[c] = rho; }
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) vstore<R, V, NT>(dst + k * geo.plane + me, outv[k]);
    if (TURB) {
        vstore<R, V, NT>(dst + K_QEQ * geo.plane + me, hq);
        vstore<R, V, NT>(dst + K_RHO * geo.plane + me, hr);
    }
}

// ------------------------------------------------------------------------------------------
// Two time steps per launch on the wall-free interior (temporal blocking through LDS).
//
// A workgroup owns a TX x TY tile of cells that are at least 4 cells away from every wall and
// slab edge.  Phase 1 performs step n -> n+1 for the tile plus a one-cell rim (rounded out to
// whole vectors in x: TX + 2V columns, TY + 2 rows), pulling from global memory, and leaves the
// post-collision populations in LDS; phase 2 performs step n+1 -> n+2 for the tile, pulling from
// LDS, and stores to global memory.  HBM traffic per cell and step drops from 18 words to about
// (9 * 1.39 + 9) / 2 = 10.8.  No wall logic is needed: every cell whose value is used is an
// ordinary cell in both semantics (MRT.py windows and kept slots only involve cells within two
// cells of a wall).  Rim cells that phase 2 does not use may be computed from ghost/pad data and
// hold garbage -- harmless.  Per-cell arithmetic is the same operation sequence as update_cell,
// so the result is bit-identical to two single steps.
// ------------------------------------------------------------------------------------------
template <typename R, int COLL, int V, bool TURB>
__device__ __forceinline__ void collide_vec(const typename VecT<R, V>::type (&in)[Q], const Relax<R>& w0,
                                            typename VecT<R, V>::type (&outv)[Q],
                                            typename VecT<R, V>::type& hq, typename VecT<R, V>::type& hr, bool wl = false, bool wr = false,
                                            int kind = 0) {
    // hq, hr: Smagorinsky history of the cells (in: previous step, out: this step); untouched unless TURB
    // The streaming kernel with the walls inside (lbm_stream.hpp) passes wall cells, their wall rules already applied to in[]:
    // wl / wr: the first / last cell is a side-wall cell (x = 0 / X - 1): u = 0 (MRT_GPU.py:396-399; as update_vec);
    // kind 1: the V cells are lid cells -- rho = (f0+f1+f3) + 2 (f2+f5+f6), u = (uLB, 0) (MRT_GPU.py:400-405); kind 2: bottom-wall
    // cells, u = 0
    if constexpr (sizeof(R) == 4 && V == 4) {
        // fp32: two cells per operation (packed math), same lane-wise IEEE operations as the scalar form below
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x2 g[Q], out[Q], q2 = f32x2(0.f);
#pragma unroll
            for (int k = 0; k < Q; ++k) g[k] = p == 0 ? in[k].xy : in[k].zw;
            f32x2 w_nu = (f32x2)(w0.w_nu);
            if (TURB) w_nu = smagorinsky_omega<f32x2, coll_is_fast(COLL)>(g, p == 0 ? hq.xy : hq.zw, p == 0 ? hr.xy : hr.zw, w0.w_nu);
            f32x2 rho = ((((((((g[0] + g[1]) + g[2]) + g[3]) + g[4]) + g[5]) + g[6]) + g[7]) + g[8]);
            if (kind == 1) rho = ((g[0] + g[1]) + g[3]) + 2.f * ((g[2] + g[5]) + g[6]);
            f32x2 ux = f32x2(0.f), uy = f32x2(0.f);
            if (!coll_is_mrt(COLL) || TURB) {
                if (kind == 0) {
                    ux = div_<coll_is_fast(COLL)>((((((g[1] - g[3]) + g[5]) - g[6]) - g[7]) + g[8]), rho);
                    uy = div_<coll_is_fast(COLL)>((((((g[2] - g[4]) + g[5]) + g[6]) - g[7]) - g[8]), rho);
                    if (wl && p == 0) { ux.x = 0.f; uy.x = 0.f; }
                    if (wr && p == 1) { ux.y = 0.f; uy.y = 0.f; }
                } else {
                    ux = f32x2(kind == 1 ? w0.uLB : 0.f); uy = f32x2(0.f);
                }
            }
            equ_collide<f32x2, COLL, TURB>(g, rho, ux, uy, w0, w_nu, out, q2);
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                if (p == 0) outv[k].xy = out[k];
                else outv[k].zw = out[k];
            }
            if (TURB) {
                if (p == 0) { hq.xy = q2; hr.xy = rho; }
                else { hq.zw = q2; hr.zw = rho; }
            }
#ifdef LBM_SEQ_HALVES
            if (p == 0) __builtin_amdgcn_sched_barrier(0);   // (experiment: the two halves one after the other -- fewer live temporaries)
#endif
        }
    } else {
#pragma unroll
        for (int c = 0; c < V; ++c) {
            R g[Q], out[Q], q2 = (R)0;
#pragma unroll
            for (int k = 0; k < Q; ++k) g[k] = in[k][c];
            R w_nu = w0.w_nu;
            if (TURB) w_nu = smagorinsky_omega<R, coll_is_fast(COLL)>(g, hq[c], hr[c], w0.w_nu);
            R rho = ((((((((g[0] + g[1]) + g[2]) + g[3]) + g[4]) + g[5]) + g[6]) + g[7]) + g[8]);
            if (kind == 1) rho = ((g[0] + g[1]) + g[3]) + (R)2. * ((g[2] + g[5]) + g[6]);
            R ux = (R)0, uy = (R)0;
            if (!coll_is_mrt(COLL) || TURB) {
                if (kind == 0) {
                    ux = div_<coll_is_fast(COLL)>((((((g[1] - g[3]) + g[5]) - g[6]) - g[7]) + g[8]), rho);
                    uy = div_<coll_is_fast(COLL)>((((((g[2] - g[4]) + g[5]) + g[6]) - g[7]) - g[8]), rho);
                    if ((wl && c == 0) || (wr && c == V - 1)) { ux = (R)0; uy = (R)0; }
                } else {
                    ux = kind == 1 ? w0.uLB : (R)0; uy = (R)0;
                }
            }
            equ_collide<R, COLL, TURB>(g, rho, ux, uy, w0, w_nu, out, q2);
#pragma unroll
            for (int k = 0; k < Q; ++k) outv[k][c] = out[k];
            if (TURB) { hq[c] = q2; hr[c] = rho; }
        }
    }
}

template <typename R, int COLL, int V, int TX, int TY, int NT, bool TURB>
__device__ __forceinline__ void update_tile2(const R* __restrict__ src, R* __restrict__ dst, const Geo& geo,
                                             const Relax<R>& w, R* __restrict__ lds, int tx0, int ty0, int xe, int ye) {
    typedef typename VecT<R, V>::type T;
    constexpr int PW = TX + 2 * V, PH = TY + 2, PVC = PW / V, TVC = TX / V;
    // phase 1: rows ty0-1 .. ty0+TY, columns tx0-V .. tx0+TX+V-1  ->  lds[Q (+2)][PH][PW]
    for (int i = threadIdx.x; i < PH * PVC; i += NT) {
        const int r = i / PVC, vc = i - r * PVC;
        const int x0 = tx0 - V + vc * V, y = ty0 - 1 + r;
        if (x0 < geo.nx && y < geo.ny) {
            T in[Q], outv[Q], hq, hr;
#pragma unroll
            for (int k = 0; k < Q; ++k)
                in[k] = vload<R, V, false>(src + k * geo.plane + geo.at(x0 - cxk(k), y + cyk(k)), cxk(k) == 0);
            if (TURB) {
                hq = vload<R, V, false>(src + K_QEQ * geo.plane + geo.at(x0, y), true);
                hr = vload<R, V, false>(src + K_RHO * geo.plane + geo.at(x0, y), true);
            }
            collide_vec<R, COLL, V, TURB>(in, w, outv, hq, hr);
#pragma unroll
            for (int k = 0; k < Q; ++k) *reinterpret_cast<T*>(lds + ((k * PH + r) * PW + vc * V)) = outv[k];
            if (TURB) {
                *reinterpret_cast<T*>(lds + ((K_QEQ * PH + r) * PW + vc * V)) = hq;
                *reinterpret_cast<T*>(lds + ((K_RHO * PH + r) * PW + vc * V)) = hr;
            }
        }
    }
    __syncthreads();
    // phase 2: the tile itself, pulling from LDS
    for (int i = threadIdx.x; i < TY * TVC; i += NT) {
        const int r = i / TVC, vc = i - r * TVC;
        const int x0 = tx0 + vc * V, y = ty0 + r;
        if (x0 >= xe || y >= ye) continue;
        T in[Q], outv[Q], hq, hr;
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            // one aligned 16-byte LDS read of the thread's own columns + ONE scalar for the x -+ 1 neighbour
            // (V scalar reads at a stride of V words are a V-way bank conflict)
            const R* p = lds + ((k * PH + (r + 1 + cyk(k))) * PW + (V + vc * V));
            const T own = *reinterpret_cast<const T*>(p);
            if (cxk(k) == 0) {
                in[k] = own;
            } else if (cxk(k) > 0) {
                in[k][0] = p[-1];
#pragma unroll
                for (int c = 1; c < V; ++c) in[k][c] = own[c - 1];
            } else {
#pragma unroll
                for (int c = 0; c < V - 1; ++c) in[k][c] = own[c + 1];
                in[k][V - 1] = p[V];
            }
        }
        if (TURB) {
            hq = *reinterpret_cast<const T*>(lds + ((K_QEQ * PH + (r + 1)) * PW + (V + vc * V)));
            hr = *reinterpret_cast<const T*>(lds + ((K_RHO * PH + (r + 1)) * PW + (V + vc * V)));
        }
        collide_vec<R, COLL, V, TURB>(in, w, outv, hq, hr);
        const long long me = geo.at(x0, y);
#pragma unroll
        for (int k = 0; k < Q; ++k) vstore<R, V, false>(dst + k * geo.plane + me, outv[k]);
        if (TURB) {
            vstore<R, V, false>(dst + K_QEQ * geo.plane + me, hq);
            vstore<R, V, false>(dst + K_RHO * geo.plane + me, hr);
        }
    }
}

// ------------------------------------------------------------------------------------------
// S time steps per launch with ONE LDS buffer updated in place (used with S = 3, the default
// without the Smagorinsky closure: profiles/r01_logs/perf11.log).  One vector cell per thread: the region is PH = TY + 2(S-1) rows
// by PW = TX + 2V columns and the workgroup has PH * PW / V threads.  Step 1 pulls from global
// memory; steps 2..S pull from LDS into registers, wait for everyone, compute, and overwrite
// their own slot; the region that is still valid shrinks by one cell per step (the V-wide rim in
// x allows up to V + 1 steps).  Threads outside the valid region compute garbage that nobody uses.
// ------------------------------------------------------------------------------------------
// LDS holds only the six directions that cross rows (k = 2, 4, 5, 6, 7, 8); the three that stay in the row are taken
// from the thread's own registers (k = 0) and from the neighbouring lanes' registers by a wave shuffle (k = 1, 3: the row of
// PVC vector cells sits in PVC consecutive lanes).  Two thirds of the LDS footprint -> three workgroups per CU.
constexpr int TB_LDS_PLANES = 6;
__host__ __device__ constexpr int lds_slot(int k) { return k == 2 ? 0 : k == 4 ? 1 : k - 3; }   // 5,6,7,8 -> 2,3,4,5

template <typename R, int COLL, int V, int TX, int TY, int S, bool TURB, int RV = 1>
__device__ __forceinline__ void update_tile_inplace(const R* __restrict__ src, R* __restrict__ dst, const Geo& geo,
                                                    const Relax<R>& w, R* __restrict__ lds, int tx0, int ty0, int xe, int ye) {
    // RV: vector cells of rim on each side in x (fp64 vectors hold two cells: more than three steps need two of them)
    typedef typename VecT<R, V>::type T;
    constexpr int PW = TX + 2 * RV * V, PH = TY + 2 * (S - 1), PVC = PW / V;
    static_assert(S - 1 <= RV * V, "the x rim is RV * V cells wide");
    static_assert(64 % PVC == 0, "a row of vector cells must not straddle two waves");
    const int r = threadIdx.x / PVC, vc = threadIdx.x % PVC;
    const int x0 = tx0 - RV * V + vc * V, y = ty0 - (S - 1) + r;
    T in[Q], outv[Q], hq, hr;
    const bool inside = x0 < geo.nx && y < geo.ny;
    if (inside) {
#pragma unroll
        for (int k = 0; k < Q; ++k)
            in[k] = vload<R, V, false>(src + k * geo.plane + geo.at(x0 - cxk(k), y + cyk(k)), cxk(k) == 0);
        if (TURB) {   // the Smagorinsky history is cell-local: it stays in this thread's registers for all S steps
            hq = vload<R, V, false>(src + K_QEQ * geo.plane + geo.at(x0, y), true);
            hr = vload<R, V, false>(src + K_RHO * geo.plane + geo.at(x0, y), true);
        }
        collide_vec<R, COLL, V, TURB>(in, w, outv, hq, hr);
    }
#pragma unroll
    for (int s = 2; s <= S; ++s) {
        if (inside) {
#pragma unroll
            for (int k = 0; k < Q; ++k)
                if (cyk(k) != 0) *reinterpret_cast<T*>(lds + ((lds_slot(k) * PH + r) * PW + vc * V)) = outv[k];
        }
        // the in-row directions never leave the registers: east-moving values come from the left neighbour's last
        // element, west-moving ones from the right neighbour's first (garbage at the two rim columns, which is fine)
        const R from_left = __shfl_up(outv[1][V - 1], 1);
        const R from_right = __shfl_down(outv[3][0], 1);
        __syncthreads();
        const bool act = r >= s - 1 && r < PH - (s - 1);
        if (act) {
            in[0] = outv[0];
            in[1][0] = from_left;
            in[3][V - 1] = from_right;
#pragma unroll
            for (int c = 1; c < V; ++c) { in[1][c] = outv[1][c - 1]; in[3][c - 1] = outv[3][c]; }
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                if (cyk(k) == 0) continue;
                const R* p = lds + ((lds_slot(k) * PH + (r + cyk(k))) * PW + vc * V);
                const T own = *reinterpret_cast<const T*>(p);
                if (cxk(k) == 0) {
                    in[k] = own;
                } else if (cxk(k) > 0) {
                    in[k][0] = p[-1];
#pragma unroll
                    for (int c = 1; c < V; ++c) in[k][c] = own[c - 1];
                } else {
#pragma unroll
                    for (int c = 0; c < V - 1; ++c) in[k][c] = own[c + 1];
                    in[k][V - 1] = p[V];
                }
            }
        }
        if (s < S) __syncthreads();   // everyone has read before anyone overwrites in place
        // (the rim columns are needed by the next step but not after the last one)
        if (act && (s < S || (vc >= RV && vc < PVC - RV))) collide_vec<R, COLL, V, TURB>(in, w, outv, hq, hr);
    }
    if (r >= S - 1 && r < PH - (S - 1) && vc >= RV && vc < PVC - RV && x0 < xe && y < ye) {
        const long long me = geo.at(x0, y);
#pragma unroll
        for (int k = 0; k < Q; ++k) vstore<R, V, false>(dst + k * geo.plane + me, outv[k]);
        if (TURB) {
            vstore<R, V, false>(dst + K_QEQ * geo.plane + me, hq);
            vstore<R, V, false>(dst + K_RHO * geo.plane + me, hr);
        }
    }
}

}  // namespace lbm
