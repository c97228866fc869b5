// lbm_kernels.hpp -- the __global__ kernels of liblbm_hip.so (templates; see lbm_device.hpp for the per-cell operators).
// Included by the host units (through lbm_host.hpp) and by lbm_tiles_f32.hip / lbm_tiles_f64.hip, which hold the explicit
// instantiations of the multi-step tile kernel so that the three translation units compile in parallel (lbm_tiles_inst.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "lbm_device.hpp"

using namespace lbm;

constexpr int BLK = 256;

// A batch of independent lattices advanced by one launch (lbm_params.batch): lattice z of the batch lives `stride` elements
// further on in every device buffer and has its own relaxation rates w[z].  w == nullptr: a single lattice, rates by value.
template <typename R>
struct Batch {
    long long stride;
    const Relax<R>* w;
};
#define LBM_BATCH_SELECT(z)            \
    if (bt.w) {                        \
        src += (z) * bt.stride;        \
        dst += (z) * bt.stride;        \
        w = bt.w[(z)];                 \
    }

// Generic fused step: one thread per cell, rows y = row0 + blockIdx.y * row_stride.
template <typename R, int COLL, int SEM, bool TURB>
__global__ __launch_bounds__(BLK) void k_step_generic(const R* __restrict__ src, R* __restrict__ dst, Geo geo,
                                                      Relax<R> w, Batch<R> bt, int raw, int row0, int row_stride) {
    const int x = blockIdx.x * BLK + threadIdx.x;
    const int y = row0 + blockIdx.y * row_stride;
    if (x >= geo.nx) return;
    LBM_BATCH_SELECT(blockIdx.z)
    update_cell<R, COLL, SEM, TURB>(src, dst, geo, w, raw, x, y);
}

// Vector fused step (MRT_GPU.py semantics): 1-D grid of nrows * nxb blocks; a block owns
// BLK * V consecutive cells of one row.  Blocks are dealt round-robin to the 8 XCDs, so block
// b is remapped such that every XCD walks its own contiguous band of rows (measured +5 % on
// the 18-stream access pattern; speed only, any placement is correct).
#ifndef LBM_VEC_MIN_WAVES
#define LBM_VEC_MIN_WAVES 1
#endif
template <typename R, int COLL, int V, bool NT, bool TURB>
__global__ __launch_bounds__(BLK, LBM_VEC_MIN_WAVES) void k_step_vec(const R* __restrict__ src, R* __restrict__ dst, Geo geo, Relax<R> w,
                                                  Batch<R> bt, int raw, int row0, int row_stride, int nxb, int nblocks) {
    LBM_BATCH_SELECT(blockIdx.y)
    int b = blockIdx.x;
    const int per = nblocks >> 3;
    if (b < (per << 3)) b = (b & 7) * per + (b >> 3);
    const int y = row0 + (b / nxb) * row_stride;
    const int gy = geo.y0 + y;
    if (gy == 0 || gy == geo.NY - 1) {
        // lid / bottom-wall row: one cell per thread and pass (consecutive threads -> consecutive cells), so that a
        // narrow lattice needs ONE pass instead of V dependent ones -- on small lattices this row is the critical path
        const int xb0 = (b % nxb) * BLK * V;
#pragma unroll 1
        for (int j = 0; j < V; ++j) {
            const int x = xb0 + j * BLK + threadIdx.x;
            if (x < geo.nx) update_cell<R, COLL, SEM_GPU, TURB>(src, dst, geo, w, raw, x, y);
        }
        return;
    }
    const int x0 = ((b % nxb) * BLK + threadIdx.x) * V;
    if (x0 >= geo.nx) return;
    update_vec<R, COLL, V, NT, TURB>(src, dst, geo, w, raw, x0, y);
}

// ---- two steps per launch (see update_tile2 in lbm_device.hpp) --------------------------------------
constexpr int TB_F = 4;      // cells within F of a wall / slab edge are advanced by single steps (8 with four steps per launch)
constexpr int TB_NT = 512;   // threads per tile
// Tile shape (measured sweep, profiles/r01_logs/tb2.log, tb3.log): wide and short wins -- 62 vectors (248 fp32 / 124 fp64 cells)
// x 6 rows: phase 1 is 8 rows x 64 vectors = exactly one vector cell per thread and one wave per 1-KiB row segment
// (9 x 8 x 256 x 4 B = 72 KiB of LDS: two tiles per CU).  The row re-reads of the short tile are served by L2.
// With the two Smagorinsky history planes: 30 vectors x 12 rows (11 x 14 x 128 x 4 B = 77 KiB).
template <bool TURB> constexpr int tb_ty() { return TURB ? 12 : 6; }
template <bool TURB> constexpr int tb_txv() { return TURB ? 30 : 62; }

template <typename R, int COLL, bool TURB>
__global__ __launch_bounds__(TB_NT) void k_step2_deep(const R* __restrict__ src, R* __restrict__ dst, Geo geo, Relax<R> w,
                                                      Batch<R> bt, int xe, int ye, int ntx, int ntiles) {
    LBM_BATCH_SELECT(blockIdx.y)
    constexpr int V = 16 / (int)sizeof(R), TX = tb_txv<TURB>() * V, TY = tb_ty<TURB>();
    __shared__ __align__(16) R lds[(TURB ? Q + 2 : Q) * (TY + 2) * (TX + 2 * V)];
    int b = blockIdx.x;
    const int per = ntiles >> 3;
    if (b < (per << 3)) b = (b & 7) * per + (b >> 3);   // every XCD walks its own band of tile rows
    update_tile2<R, COLL, V, TX, TY, TB_NT, TURB>(src, dst, geo, w, lds, TB_F + (b % ntx) * TX, TB_F + (b / ntx) * TY, xe, ye);
}

// All S frame passes of a multi-step in ONE launch.  The frame (width F) is cut into rectangles: segments of FR_L columns of the
// top and bottom row strips, segments of FR_L rows of the left and right column strips; one workgroup per rectangle.  Pass i
// (margin m = S - i) computes the rectangle grown by m cells in x and y (clipped to the lattice; the rows of a slab interface
// grow m rows into the neighbour's rows of the deep halo), all of it inside the frame of width F + m, from the output of pass
// i - 1 -- every cell a workgroup reads it has written itself one pass earlier, so workgroups never wait for each other; where
// grown rectangles overlap, both workgroups write the same bits.  Each pass has its own scratch lattice (a ping-pong pair would
// let a fast workgroup overwrite cells a slow neighbour still reads); the last pass writes lat[b].  Same per-cell operations as
// S launches of k_step_frame.
template <typename R>
struct FramePtrs {
    const R* src;    // state n
    R* pass[8];      // output of pass 1 .. S (pass[S - 1] = the destination lattice)
};

template <typename R>
__device__ __forceinline__ R* pass_ptr(const FramePtrs<R>& fp, int j) {   // (selects, not an indexed load: no private-memory copy)
    return j == 0 ? fp.pass[0] : j == 1 ? fp.pass[1] : j == 2 ? fp.pass[2] : j == 3 ? fp.pass[3] : j == 4 ? fp.pass[4] : j == 5 ? fp.pass[5]
                                                                                                 : j == 6 ? fp.pass[6] : fp.pass[7];
}

// LDS bytes a workgroup of the fused frame passes may use (= the tile kernel's buffer)
constexpr int FRAME_LDS_BYTES = 48 * 1024;
// ... and inside the tile launch of a lone lattice: two workgroups of 512 threads fill a CU's wave slots at <= 128 VGPRs, so each
// may take up to half of the 160 KiB of LDS at no cost in occupancy; the windows of 64-cell segments (fp32, five passes: 73 KiB)
// then fit and the frame chain -- the critical path of small and medium lattices -- stops going through the scratch lattices
constexpr int TILE_FRAME_LDS_BYTES = 76 * 1024;

template <typename R, int COLL, int SEM, bool TURB, int NT>
__device__ __forceinline__ void frame_passes(const FramePtrs<R>& fp, long long boff, const Geo& geo, const Relax<R>& w, int F, int S,
                                             int nsegx, int nsegy, int lo, int hi, int b, int FR_L, R* lds, int bands = 0) {
    // lo / hi: 0 = no neighbour on that side; e + 1 = a neighbour whose rows lie in the ghost rows, and the row strips own e of
    // them (e = 0: a launch unit; e = 1: the recomputation of the lattice of the step before the last, whose first ghost rows
    // the field export pulls from)
    const int elo = lo > 0 ? lo - 1 : 0, ehi = hi > 0 ? hi - 1 : 0;
    // bands (bit 0: rows below F, bit 1: rows from ny - F): the streaming kernel computes that row strip between the column
    // strips itself (as a short segment, k_stream), so the strip has no workgroups here and the column strips run to the slab's
    // edge (and e rows beyond) on that side
    int x0, x1, y0, y1;   // owned rectangle [x0, x1) x [y0, y1)
    if (b < 2 * nsegx) {
        const int seg = b % nsegx;
        x0 = seg * FR_L; x1 = min(geo.nx, x0 + FR_L);
        if (b < nsegx) { if (bands & 1) return; y0 = -elo; y1 = F; }
        else { if (bands & 2) return; y0 = geo.ny - F; y1 = geo.ny + ehi; }
    } else {
        b -= 2 * nsegx;
        const int seg = b % nsegy;
        const int ybeg = (bands & 1) ? -elo : F, yend = (bands & 2) ? geo.ny + ehi : geo.ny - F;
        y0 = ybeg + seg * FR_L; y1 = min(yend, y0 + FR_L);
        x0 = b < nsegy ? 0 : geo.nx - F; x1 = x0 + F;
    }
    if (lds) {
        // The output of the intermediate passes stays in LDS: a window = the rectangle of pass 1 plus a ring of one cell (ghost
        // positions where wall cells park kept slots and densities), two buffers in turn.  Pass 1 reads the lattice, the last
        // pass writes the lattice; every window entry a pass reads was written by the pass before (or is overwritten by a wall
        // rule before use); the window is zeroed first so that such dead reads see numbers.
        constexpr int NP = TURB ? Q + 2 : Q;
        const int m1 = S - 1;
        const int xa1 = max(0, x0 - m1), xb1 = min(geo.nx, x1 + m1);
        const int ya1 = max(lo ? -(m1 + elo) : 0, y0 - m1), yb1 = min(geo.ny + (hi ? m1 + ehi : 0), y1 + m1);
        Window win;
        win.x0 = xa1 - 1; win.y0 = ya1 - 1; win.pitch = xb1 - xa1 + 2; win.plane = win.pitch * (yb1 - ya1 + 2);
        R* const buf0 = lds;
        R* const buf1 = lds + NP * win.plane;
        for (int t = threadIdx.x; t < 2 * NP * win.plane; t += NT) lds[t] = (R)0;
        __syncthreads();
        for (int i = 1; i <= S; ++i) {
            const int m = S - i;
            const int xa = max(0, x0 - m), xb = min(geo.nx, x1 + m);
            const int ya = max(lo ? -(m + elo) : 0, y0 - m), yb = min(geo.ny + (hi ? m + ehi : 0), y1 + m);
            const int wx = xb - xa, n = wx * (yb - ya);
            R* const wr = (i & 1) ? buf0 : buf1;
            const R* const rd = (i & 1) ? buf1 : buf0;
            if (i == 1) {
                const R* src = fp.src + boff;
                for (int t = threadIdx.x; t < n; t += NT)
                    update_cell_a<R, COLL, SEM, TURB, Geo, Window>(src, geo, wr, win, geo, w, 0, xa + t % wx, ya + t / wx);
            } else if (i < S) {
                for (int t = threadIdx.x; t < n; t += NT)
                    update_cell_a<R, COLL, SEM, TURB, Window, Window>(rd, win, wr, win, geo, w, 0, xa + t % wx, ya + t / wx);
            } else {
                R* dst = pass_ptr(fp, S - 1) + boff;
                for (int t = threadIdx.x; t < n; t += NT)
                    update_cell_a<R, COLL, SEM, TURB, Window, Geo>(rd, win, dst, geo, geo, w, 0, xa + t % wx, ya + t / wx);
            }
            __syncthreads();
        }
        return;
    }
    for (int i = 1; i <= S; ++i) {
        const int m = S - i;
        const int xa = max(0, x0 - m), xb = min(geo.nx, x1 + m);
        const int ya = max(lo ? -(m + elo) : 0, y0 - m), yb = min(geo.ny + (hi ? m + ehi : 0), y1 + m);
        const int wx = xb - xa, n = wx * (yb - ya);
        const R* src = (i == 1 ? fp.src : pass_ptr(fp, i - 2)) + boff;
        R* dst = pass_ptr(fp, i - 1) + boff;
        for (int t = threadIdx.x; t < n; t += NT) update_cell<R, COLL, SEM, TURB>(src, dst, geo, w, 0, xa + t % wx, ya + t / wx);
        __syncthreads();   // workgroup-scope release / acquire: the next pass reads what this one wrote (global memory)
    }
}

// NT threads per workgroup: 256, or 1024 for the passes of a slab that go through the scratch lattices (a pass is then one sweep of
// the workgroup instead of four or five dependent round trips to L2: the frame kernel is on the critical path exchange -> frame
// -> exchange of a slab, profiles/r02_logs/slab_loopback2.log)
template <typename R, int COLL, int SEM, bool TURB, int NT>
__global__ __launch_bounds__(NT) void k_frame_multi(FramePtrs<R> fp, Geo geo, Relax<R> w, Batch<R> bt, int F, int S, int nsegx, int nsegy,
                                                    int lo, int hi, int seg, int use_lds) {
    __shared__ __align__(16) R lds[FRAME_LDS_BYTES / sizeof(R)];
    long long boff = 0;
    if (bt.w) { boff = (long long)blockIdx.y * bt.stride; w = bt.w[blockIdx.y]; }
    frame_passes<R, COLL, SEM, TURB, NT>(fp, boff, geo, w, F, S, nsegx, nsegy, lo, hi, (int)blockIdx.x, seg, use_lds ? lds : nullptr);
}

// The same passes for the wall frame of a lone lattice whose bulk runs in the streaming kernel, launched on the second stream
// to run BESIDE it: a streaming workgroup takes all of a CU's LDS and four waves per SIMD x 96 .. 128 VGPRs, which can leave room
// for one wave per SIMD of a kernel without LDS (the passes go through the scratch lattices) and with 42 .. 91 VGPRs; lbm_create
// compares the register counts of the two kernel variants (hipFuncGetAttributes) and chooses.
template <typename R, int COLL, int SEM, bool TURB>
__global__ __launch_bounds__(BLK) void k_frame_beside(FramePtrs<R> fp, Geo geo, Relax<R> w, int F, int S, int nsegx, int nsegy, int seg) {
    frame_passes<R, COLL, SEM, TURB, BLK>(fp, 0, geo, w, F, S, nsegx, nsegy, 0, 0, (int)blockIdx.x, seg, nullptr);
}

// S = 3 .. 5 steps per launch: region of 512 vector cells = one per thread, 48 KiB of LDS.  The
// occupancy floor of 4 waves per SIMD (<= 128 VGPRs) keeps two workgroups on a CU: the MRT / TRT + Smagorinsky variants would
// otherwise take 132 and run one (perf22.log vs perf23.log: fp32 MRT turb 96 -> 115 GLUPS).  WIDE: region 32 vectors x 16 rows;
// otherwise 16 vectors x 32 rows (less rim work, shorter row segments).  The tile is the region minus the rim: V cells
// left and right, S - 1 rows above and below.  F = frame width (4; 8 for S = 4 in fp32).
template <typename R, int COLL, int SEM, int S, bool WIDE, bool TURB>
__global__ __launch_bounds__(512, 4) void k_stepS_deep(const R* __restrict__ src, R* __restrict__ dst, Geo geo, Relax<R> w,
                                                    Batch<R> bt, int F, int xe, int ye, int ntx, int ntiles, FramePtrs<R> fp, int nframe,
                                                    int nsegx, int nsegy, int seg, int use_lds) {
    constexpr int V = 16 / (int)sizeof(R), PVC = WIDE ? 32 : 16, PH = 512 / PVC;
    constexpr int RV = (S - 1 + V - 1) / V;        // rim vectors per side: 1, or 2 for fp64 beyond three steps
    constexpr int TX = (PVC - 2 * RV) * V, TY = PH - 2 * (S - 1), PW = TX + 2 * RV * V;
    constexpr int TILE_ELEMS = TB_LDS_PLANES * PH * PW + 2 * V, FRAME_ELEMS = TILE_FRAME_LDS_BYTES / (int)sizeof(R);
    __shared__ __align__(16) R lds_raw[TILE_ELEMS > FRAME_ELEMS ? TILE_ELEMS : FRAME_ELEMS];   // one vector of slack at each end: rim columns
    if ((int)blockIdx.x < nframe) {                                         // read one element past their row
        // A lone lattice: the first nframe workgroups of the launch do the S frame passes (frame_passes), the rest the tiles --
        // one launch per S steps and no cross-stream dependency (between slabs the frame stays a launch of its own on the
        // communication stream: nframe = 0).  The frame workgroups run S dependent passes and take the longest: they go first.
        long long boff = 0;
        if (bt.w) { boff = (long long)blockIdx.y * bt.stride; w = bt.w[blockIdx.y]; }
        static_assert(sizeof(lds_raw) >= TILE_FRAME_LDS_BYTES, "frame window buffer");
        frame_passes<R, COLL, SEM, TURB, 512>(fp, boff, geo, w, F, S, nsegx, nsegy, 0, 0, (int)blockIdx.x, seg, use_lds ? lds_raw : nullptr);
        return;
    }
    LBM_BATCH_SELECT(blockIdx.y)
    int b = blockIdx.x - nframe;
    const int per = ntiles >> 3;
    if (b < (per << 3)) b = (b & 7) * per + (b >> 3);
    update_tile_inplace<R, COLL, V, TX, TY, S, TURB, RV>(src, dst, geo, w, lds_raw + V, F + (b % ntx) * TX, F + (b / ntx) * TY, xe, ye);
}

// One single step on the frame of width W around the slab: rows [0, W) and [ny-W, ny) in full, columns [0, W) and
// [nx-W, nx) of the rows in between.  One thread per cell, complete wall / kept-slot logic.
// elo / ehi: the row strips start elo rows above row 0 / end ehi rows below row ny - 1, in the ghost rows that hold the
// neighbour slab's rows (deep halo: the passes of a multi-step recompute a shrinking band of the neighbour's rows instead of
// exchanging one row per pass).
template <typename R, int COLL, int SEM, bool TURB>
__global__ __launch_bounds__(BLK) void k_step_frame(const R* __restrict__ src, R* __restrict__ dst, Geo geo, Relax<R> w, Batch<R> bt, int W, int elo, int ehi,
                                                    int vec_rows) {
    LBM_BATCH_SELECT(blockIdx.y)
    long long t = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long ncol = 2LL * W * (geo.ny - 2 * W);
    int x, y;
    if (SEM == SEM_GPU && vec_rows) {
        // row strips by vector cells (16 B per access, as k_step_vec: side walls in-line, lid / bottom-wall rows cell by cell)
        constexpr int V = 16 / (int)sizeof(R);
        const int nvx = geo.nx / V;
        const long long ntop = (long long)(W + elo) * nvx, nbot = (long long)(W + ehi) * nvx;
        if (t < ntop + nbot) {
            const int o = (int)(t < ntop ? t : t - ntop);
            const int x0 = (o % nvx) * V, yv = (t < ntop ? -elo : geo.ny - W) + o / nvx, gy = geo.y0 + yv;
            if (gy == 0 || gy == geo.NY - 1) {
#pragma unroll 1
                for (int c = 0; c < V; ++c) update_cell<R, COLL, SEM_GPU, TURB>(src, dst, geo, w, 0, x0 + c, yv);
            } else {
                update_vec<R, COLL, V, false, TURB>(src, dst, geo, w, 0, x0, yv);
            }
            return;
        }
        t -= ntop + nbot;
    } else {
        const long long ntop = (long long)(W + elo) * geo.nx, nbot = (long long)(W + ehi) * geo.nx;
        if (t < ntop + nbot) {
            const int o = (int)(t < ntop ? t : t - ntop);
            update_cell<R, COLL, SEM, TURB>(src, dst, geo, w, 0, o % geo.nx, (t < ntop ? -elo : geo.ny - W) + o / geo.nx);
            return;
        }
        t -= ntop + nbot;
    }
    if (t >= ncol) return;
    const long long half = (long long)W * (geo.ny - 2 * W);
    const int strip = (int)(t / half), o = (int)(t % half);
    y = W + o / W;
    x = (strip == 0 ? 0 : geo.nx - W) + o % W;
    update_cell<R, COLL, SEM, TURB>(src, dst, geo, w, 0, x, y);
}

// ---- the reference's own scheme, for A/B against the fused pull kernels (LBM_KERNEL_PUSH) ---------------------------------------
// Two launches per step, as funRT + funBC (MRT_GPU.py:339-420,536-660 / 665-698): `fin` holds the populations a step starts
// from; k_push_collide relaxes every cell and PUSHES slot k into the persistent array `ftemp` at (x + cx, y - cy) if that cell
// streams it (slots outside their window keep what ftemp held); k_push_bc applies the wall rules to ftemp in place, with the
// equilibrium of the state the step started from, and copies ftemp into the next `fin`.  About 49 words per cell and step
// against the 18 of the fused pull kernel.  Same per-cell operators, so the results are bit-identical to the pull kernels.
template <typename R, int COLL, int SEM>
__global__ __launch_bounds__(BLK) void k_push_collide(const R* __restrict__ fin, R* __restrict__ ftemp, Geo geo, Relax<R> w) {
    const int x = blockIdx.x * BLK + threadIdx.x, y = blockIdx.y;
    if (x >= geo.nx) return;
    const int X = geo.nx, Y = geo.NY, gy = geo.y0 + y;
    R g[Q], rho, ux, uy, out[Q], q2;
    const long long me = geo.at(x, y);
#pragma unroll
    for (int k = 0; k < Q; ++k) g[k] = fin[k * geo.plane + me];
    macros<R, coll_is_fast(COLL)>(g, x, gy, X, Y, w.uLB, rho, ux, uy);
    equ_collide<R, COLL, false>(g, rho, ux, uy, w, w.w_nu, out, q2);
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int dx = x + cxk(k), dgy = gy - cyk(k);
        if (dx >= 0 && dx < X && dgy >= 0 && dgy < Y && in_window<SEM>(k, dx, dgy, X, Y))
            ftemp[k * geo.plane + geo.at(dx, y - cyk(k))] = out[k];
    }
}

template <typename R, int COLL, int SEM>
__global__ __launch_bounds__(BLK) void k_push_bc(const R* __restrict__ fin_old, R* __restrict__ ftemp, R* __restrict__ fin_new, Geo geo, R uLB) {
    const int x = blockIdx.x * BLK + threadIdx.x, y = blockIdx.y;
    if (x >= geo.nx) return;
    const int X = geo.nx, Y = geo.NY, gy = geo.y0 + y;
    const long long me = geo.at(x, y);
    R g[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) g[k] = ftemp[k * geo.plane + me];
    if (x == 0 || x == X - 1 || gy == 0 || gy == Y - 1) {
        R f0[Q], rho, ux, uy, fe[Q];   // equilibrium of the state this step started from (feq_g written by funRT)
#pragma unroll
        for (int k = 0; k < Q; ++k) f0[k] = fin_old[k * geo.plane + me];
        macros<R, coll_is_fast(COLL)>(f0, x, gy, X, Y, uLB, rho, ux, uy);
        equ<R>(rho, ux, uy, fe);
        wall_rules<R, SEM>(g, fe, x, gy, X, Y);
#pragma unroll
        for (int k = 0; k < Q; ++k) ftemp[k * geo.plane + me] = g[k];
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) fin_new[k * geo.plane + me] = g[k];
}

// init: raw populations = equ(rho = 1, u = (uLB on the global lid row, 0))  (MRT.py:260-268)
template <typename R>
__global__ __launch_bounds__(BLK) void k_init(R* __restrict__ lat, Geo geo, R uLB, int turb, long long bstride) {
    const int x = blockIdx.x * BLK + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= geo.nx) return;
    lat += blockIdx.z * bstride;
    R fe[Q];
    equ<R>((R)1, (geo.y0 + y) == 0 ? uLB : (R)0, (R)0, fe);
#pragma unroll
    for (int k = 0; k < Q; ++k) lat[k * geo.plane + geo.at(x, y)] = fe[k];
    if (turb) {   // Smagorinsky history: feq_g starts as a copy of fin, rho_g as 1 (MRT_GPU.py:324-326)
        lat[K_QEQ * geo.plane + geo.at(x, y)] = diag_flux<R>(fe);
        lat[K_RHO * geo.plane + geo.at(x, y)] = (R)1;
    }
}

// ---- host layout <-> lattice ---------------------------------------------------------------------------------------------
// The staging buffers hold the reference's host layout ([plane][nx][ny_local], y fastest); the lattice is x fastest.  Each
// workgroup moves a tile of TRX columns x 32 rows through LDS, so that both sides are accessed along their fast axis.
template <typename R> constexpr int trx() { return sizeof(R) == 4 ? 32 : 16; }   // 9 planes x TRX x 33 reals = 38 KiB of LDS

// staging -> raw lattice
template <typename R>
__global__ __launch_bounds__(BLK) void k_import(const R* __restrict__ stage, R* __restrict__ lat, Geo geo, R uLB, int turb, long long bstride) {
    constexpr int TRX = trx<R>(), RY = BLK / TRX;
    __shared__ R t[Q][TRX][33];
    const long long n = (long long)geo.nx * geo.ny;
    lat += blockIdx.z * bstride;
    stage += blockIdx.z * (Q * n);
    const int x0 = blockIdx.x * TRX, y0 = blockIdx.y * 32;
    {   // read along y
        const int yy = threadIdx.x & 31, xb = threadIdx.x >> 5;
        for (int xx = xb; xx < TRX; xx += BLK / 32)
            if (x0 + xx < geo.nx && y0 + yy < geo.ny) {
#pragma unroll
                for (int k = 0; k < Q; ++k) t[k][xx][yy] = stage[k * n + (long long)(x0 + xx) * geo.ny + y0 + yy];
            }
    }
    __syncthreads();
    const int tx = threadIdx.x % TRX, x = x0 + tx;
    for (int ty = threadIdx.x / TRX; ty < 32; ty += RY) {   // write along x
        const int y = y0 + ty;
        if (x >= geo.nx || y >= geo.ny) continue;
        R g[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            g[k] = t[k][tx][ty];
            lat[k * geo.plane + geo.at(x, y)] = g[k];
        }
        if (turb) {   // history := equilibrium / density of the uploaded state (there is no "previous step")
            R rho, ux, uy, fe[Q];
            macros<R>(g, x, geo.y0 + y, geo.nx, geo.NY, uLB, rho, ux, uy);
            equ<R>(rho, ux, uy, fe);
            lat[K_QEQ * geo.plane + geo.at(x, y)] = diag_flux<R>(fe);
            lat[K_RHO * geo.plane + geo.at(x, y)] = rho;
        }
    }
}

// lattice -> staging: current populations (post stream + wall rules) in host layout
template <typename R, int SEM>
__global__ __launch_bounds__(BLK) void k_export_fin(const R* __restrict__ src, Geo geo, int raw, R uLB,
                                                    R* __restrict__ stage, long long bstride) {
    constexpr int TRX = trx<R>(), RY = BLK / TRX;
    __shared__ R t[Q][TRX][33];
    const long long n = (long long)geo.nx * geo.ny;
    src += blockIdx.z * bstride;
    stage += blockIdx.z * (Q * n);
    const int x0 = blockIdx.x * TRX, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x % TRX, x = x0 + tx;
    for (int ty = threadIdx.x / TRX; ty < 32; ty += RY) {   // gather along x
        const int y = y0 + ty;
        if (x >= geo.nx || y >= geo.ny) continue;
        R g[Q];
        gather<R, SEM>(src, geo, raw, uLB, x, y, g);
#pragma unroll
        for (int k = 0; k < Q; ++k) t[k][tx][ty] = g[k];
    }
    __syncthreads();
    const int yy = threadIdx.x & 31, xb = threadIdx.x >> 5;
    for (int xx = xb; xx < TRX; xx += BLK / 32)              // write along y
        if (x0 + xx < geo.nx && y0 + yy < geo.ny) {
#pragma unroll
            for (int k = 0; k < Q; ++k) stage[k * n + (long long)(x0 + xx) * geo.ny + y0 + yy] = t[k][xx][yy];
        }
}

// lattice -> staging: macroscopic fields (with wall overrides) of the populations gathered
// from `src`; stage = [ux | uy | rho], each [nx][ny_local]
template <typename R, int SEM>
__global__ __launch_bounds__(BLK) void k_export_macro(const R* __restrict__ src, Geo geo, int raw, R uLB,
                                                      R* __restrict__ stage, long long bstride) {
    constexpr int TRX = trx<R>(), RY = BLK / TRX;
    __shared__ R t[3][TRX][33];
    const long long n = (long long)geo.nx * geo.ny;
    src += blockIdx.z * bstride;
    stage += blockIdx.z * (3 * n);
    const int x0 = blockIdx.x * TRX, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x % TRX, x = x0 + tx;
    for (int ty = threadIdx.x / TRX; ty < 32; ty += RY) {
        const int y = y0 + ty;
        if (x >= geo.nx || y >= geo.ny) continue;
        R g[Q], rho, ux, uy;
        gather<R, SEM>(src, geo, raw, uLB, x, y, g);
        macros<R>(g, x, geo.y0 + y, geo.nx, geo.NY, uLB, rho, ux, uy);
        t[0][tx][ty] = ux; t[1][tx][ty] = uy; t[2][tx][ty] = rho;
    }
    __syncthreads();
    const int yy = threadIdx.x & 31, xb = threadIdx.x >> 5;
    for (int xx = xb; xx < TRX; xx += BLK / 32)
        if (x0 + xx < geo.nx && y0 + yy < geo.ny) {
            const long long o = (long long)(x0 + xx) * geo.ny + y0 + yy;
            stage[o] = t[0][xx][yy];
            stage[n + o] = t[1][xx][yy];
            stage[2 * n + o] = t[2][xx][yy];
        }
}

// lattice -> staging: relaxation time tau + tau_turbulent of the iteration that starts from the populations gathered from `src`
// (taus_g, MRT_GPU.py:385-387); turb = 0: the constant 1 / omega.  stage = [nx][ny_local]
template <typename R, bool FAST>
__global__ __launch_bounds__(BLK) void k_export_tau(const R* __restrict__ src, Geo geo, int raw, Relax<R> w, Batch<R> bt, int turb,
                                                    R* __restrict__ stage) {
    constexpr int TRX = trx<R>(), RY = BLK / TRX;
    __shared__ R t[TRX][33];
    const long long n = (long long)geo.nx * geo.ny;
    if (bt.w) { src += blockIdx.z * bt.stride; w = bt.w[blockIdx.z]; }
    stage += blockIdx.z * n;
    const int x0 = blockIdx.x * TRX, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x % TRX, x = x0 + tx;
    for (int ty = threadIdx.x / TRX; ty < 32; ty += RY) {
        const int y = y0 + ty;
        if (x >= geo.nx || y >= geo.ny) continue;
        R tau = (R)1.0 / w.w_nu;
        if (turb) {
            R g[Q];
            gather<R, SEM_GPU>(src, geo, raw, w.uLB, x, y, g);
            const long long me = geo.at(x, y);
            tau = smagorinsky_tau<R, FAST>(g, src[K_QEQ * geo.plane + me], src[K_RHO * geo.plane + me], w.w_nu);
        }
        t[tx][ty] = tau;
    }
    __syncthreads();
    const int yy = threadIdx.x & 31, xb = threadIdx.x >> 5;
    for (int xx = xb; xx < TRX; xx += BLK / 32)
        if (x0 + xx < geo.nx && y0 + yy < geo.ny) stage[(long long)(x0 + xx) * geo.ny + y0 + yy] = t[xx][yy];
}

// Sum over the slab of ux + uy of the macroscopic state gathered from `src` (what k_export_macro would write), in double:
// partial[blockIdx.z * gridDim.x + blockIdx.x] = the block's sum; k_reduce_final adds the partial sums of each lattice in
// index order -- a fixed summation tree, so the result does not vary from run to run.
template <typename R, int SEM>
__global__ __launch_bounds__(BLK) void k_reduce_u(const R* __restrict__ src, Geo geo, int raw, R uLB, long long bstride,
                                                  double* __restrict__ partial) {
    __shared__ double red[BLK];
    src += blockIdx.z * bstride;
    const long long n = (long long)geo.nx * geo.ny;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += (long long)gridDim.x * BLK) {
        const int y = (int)(i / geo.nx), x = (int)(i - (long long)y * geo.nx);
        R g[Q], rho, ux, uy;
        gather<R, SEM>(src, geo, raw, uLB, x, y, g);
        macros<R>(g, x, geo.y0 + y, geo.nx, geo.NY, uLB, rho, ux, uy);
        acc += (double)ux + (double)uy;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = BLK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)blockIdx.z * gridDim.x + blockIdx.x] = red[0];
}

