// lbm_stream.hpp -- the large-lattice multi-step kernel: S time steps per launch, streamed down column strips.
//
// Why a second multi-step kernel: k_stepS_deep (lbm_kernels.hpp) cuts the interior into 2-D tiles; every tile re-reads and
// re-computes a rim of S - 1 cells on all four sides (S = 5, fp32: 1.5 x the reads, 1.33 x the arithmetic; fp64 1.8 x), and a
// workgroup alternates between a load phase, S compute phases and a store phase that only the second workgroup of the CU
// overlaps.  Here a workgroup owns a column STRIP (one wave-row wide: 64 lanes x V cells = 256 fp32 / 128 fp64 cells) of a
// tall segment of rows and marches down it; the rim exists in x only (R cells per side), rows are loaded once and stored once,
// and loading, the S levels and storing are all in flight at the same time in different waves.
//
// Data flow.  Row y of the strip is a BLOCK held by one wave in registers (9 planes x V cells per lane -- the register file,
// not LDS, is the big on-chip store: 512 KB per CU).  A block takes S updates; update l -> l + 1 of row y needs rows y - 1 and
// y + 1 at level l.  Blocks are skewed in time: block b performs its update number l in iteration b + 2 l, so its neighbours
// b - 1 and b + 1 have reached level l one iteration earlier and have not yet moved on by more than one level.  Neighbouring
// rows talk through LDS: after an update a block posts the three planes that move up (cy = +1: k = 2, 5, 6; read by block
// b - 1) and the three that move down (k = 4, 7, 8; read by block b + 1).  The upward post is consumed before the next post
// overwrites it; the downward one is consumed one iteration after the next post, so it alternates between two buffers.  The
// three planes that stay in the row (k = 0, 1, 3) never leave the registers: x -+ 1 neighbours come from the adjacent lane by
// DPP (v_mov_b32 wave_shr / wave_shl, 4 cycles, no LDS).  The first update of a block pulls straight from the lattice (the nine
// shifted loads of a single step), the last one stores to the lattice, and the wave then loads its next block (16 rows further
// down): 2 S = 16 blocks are in flight per workgroup (16 waves), half of them updating in any one iteration -- two per SIMD.
// ONE workgroup barrier per iteration; every LDS slot is written in even and read in odd iterations of its owner (or vice
// versa), so no second barrier is needed.
//
// Per-cell arithmetic is collide_vec, the same operation sequence as every other kernel: results are bit-identical.
// Lead-in: the first / last S - 1 rows of a segment run through the pipeline too; their higher levels are computed from rows
// that are not there (stale or uninitialised LDS, possibly NaN).  Why that garbage never reaches a stored row: the pipeline holds
// the rows [y_first, y_end) = [ya - (S - 1), yb + (S - 1)).  Level 1 of every one of them is exact (pulled from the lattice, whose
// ghost rows make every address valid).  Level l of row y is computed from level l - 1 of the rows y - 1, y, y + 1 only, so by
// induction level l is exact on [y_first + (l - 1), y_end - (l - 1)): the exact range loses ONE row per level and side, and level S
// is exact on [ya, yb) -- precisely the rows that are stored (the store tests ya <= y < yb).  The rows outside read posts of blocks
// that are not in the pipeline; what they compute from them is posted only to their own neighbours, i.e. again outside the exact
// range of the next level.  The same argument in x: a lane shift brings in one cell of garbage per level at a strip edge (DPP keeps
// the own value in lane 0 / 63), and the stored columns lie R >= S cells inside it.  (With the walls inside, a wall row or column
// ends the dependence on that side -- the wall rule replaces what would come from beyond -- so no lead rows are needed there.)
// S is a run-time argument (the loop body is the same for every level): one instantiation serves every unit length.
#pragma once
#include "lbm_kernels.hpp"

constexpr int ST_WAVES = 16;            // waves per workgroup = blocks in flight
constexpr int ST_NT = ST_WAVES * 64;
constexpr int ST_MAX_S = ST_WAVES / 2;  // 8 steps per launch at most
// LDS per wave: 3 upward planes + 2 x 3 downward planes, one wave-row (64 lanes x 16 B = 1 KiB) each
constexpr int ST_LDS_BYTES = ST_WAVES * 9 * 1024;   // 144 KiB: one workgroup per CU

// value of the lane below / above in the wave (lane 0 / 63 keep their own): one DPP move per 32-bit register
__device__ __forceinline__ int dpp_from_lower(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1
__device__ __forceinline__ int dpp_from_upper(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }   // wave_shl:1
__device__ __forceinline__ float lane_lower(float v) { return __builtin_bit_cast(float, dpp_from_lower(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float lane_upper(float v) { return __builtin_bit_cast(float, dpp_from_upper(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ double lane_lower(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)dpp_from_lower((int)(unsigned)b), hi = (unsigned)dpp_from_lower((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double lane_upper(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)dpp_from_upper((int)(unsigned)b), hi = (unsigned)dpp_from_upper((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

// ... the same with a value of its own for the lane that has no lower / upper lane (lane 0 / 63): with the walls inside the streaming
// kernel that lane holds the side-wall cell, and what the wall rule puts in place of the population from beyond the wall comes in
// through the `old` operand of the same DPP move -- no select, no branch
__device__ __forceinline__ float lane_lower_e(float v, float edge) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_upper_e(float v, float edge) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_lower_e(double v, double edge) {
    const long long b = __builtin_bit_cast(long long, v), e = __builtin_bit_cast(long long, edge);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)e, (int)(unsigned)b, 0x138, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(e >> 32), (int)(unsigned)(b >> 32), 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double lane_upper_e(double v, double edge) {
    const long long b = __builtin_bit_cast(long long, v), e = __builtin_bit_cast(long long, edge);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)e, (int)(unsigned)b, 0x130, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(e >> 32), (int)(unsigned)(b >> 32), 0x130, 0xf, 0xf, false);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
template <typename R, int V>
__device__ __forceinline__ typename VecT<R, V>::type shift_from_lower_e(typename VecT<R, V>::type own, R edge) {
    typename VecT<R, V>::type r;
    r[0] = lane_lower_e(own[V - 1], edge);
#pragma unroll
    for (int c = 1; c < V; ++c) r[c] = own[c - 1];
    return r;
}
template <typename R, int V>
__device__ __forceinline__ typename VecT<R, V>::type shift_from_upper_e(typename VecT<R, V>::type own, R edge) {
    typename VecT<R, V>::type r;
#pragma unroll
    for (int c = 0; c < V - 1; ++c) r[c] = own[c + 1];
    r[V - 1] = lane_upper_e(own[0], edge);
    return r;
}

// in[c] = own[c - 1] with own[-1] = the lower lane's own[V - 1]   (values that move towards +x)
template <typename R, int V>
__device__ __forceinline__ typename VecT<R, V>::type shift_from_lower(typename VecT<R, V>::type own) {
    typename VecT<R, V>::type r;
    r[0] = lane_lower(own[V - 1]);
#pragma unroll
    for (int c = 1; c < V; ++c) r[c] = own[c - 1];
    return r;
}
// in[c] = own[c + 1] with own[V] = the upper lane's own[0]        (values that move towards -x)
template <typename R, int V>
__device__ __forceinline__ typename VecT<R, V>::type shift_from_upper(typename VecT<R, V>::type own) {
    typename VecT<R, V>::type r;
#pragma unroll
    for (int c = 0; c < V - 1; ++c) r[c] = own[c + 1];
    r[V - 1] = lane_upper(own[0]);
    return r;
}

// rim cells per side of a strip for S steps (a multiple of the vector width)
__host__ __device__ constexpr int stream_rim(int S, int V) { return (S + V - 1) / V * V; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL loads and stores
// (s_waitcnt vmcnt(0)): here that would put the HBM round trip of the one wave that has just stored its row and prefetched the
// next on the critical path of all sixteen, every iteration (measured: 2.4 us per iteration instead of 0.7).  The prefetched
// registers are waited for where they are used (the compiler's own s_waitcnt vmcnt before the first update of the block).
#ifdef LBM_EXPERIMENT_NO_BARRIER   // timing experiment only (tools/probes): the waves run free, the results are garbage
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// ---- the walls inside the streaming kernel (k_stream_walls: a lone lattice in MRT_GPU.py semantics) --------------------------------
// Without them the cells within F of a wall are a FRAME advanced by S dependent single-step passes of the general one-cell
// update (frame_passes): 1 % of the updates, 11 % of the launch at 4096^2 and S = 8, and growing with S^2.  Here the strips span the
// whole width and the segments the whole height, and the wall cells are updated in the pipeline itself:
//   * side walls (x = 0 / X - 1, not on the lid / bottom row): MRT_GPU.py:674-682 for a resting wall is  f_a = 0 + f_b  (fe_a - fe_b
//     is exactly +0: same weight, u = 0) on the cell's pulled populations, and u = 0 in the equilibrium -- as update_vec;
//   * the lid row and the bottom row are blocks of the pipeline like any other: wall_row_rules below applies the wall rules (x rule
//     on a corner cell first, then the y rule: MRT_GPU.py:674-692) to the pulled populations, and collide_vec takes the macroscopic
//     overrides of the row (MRT_GPU.py:396-405) before the same collide<> as every cell.  The lid rule needs the equilibrium of the cell's PREVIOUS
//     macroscopic state, i.e. its previous density (u is prescribed): in the lattice it is parked in slot 0 of the ghost row
//     (wall_rho_at), in the pipeline it is carried from level to level in the LDS slots of the row's own upward post -- no row
//     reads the lid row's upward post (the bottom row's downward post), so those slots are free: no extra registers;
//   * each corner cell has ONE slot whose value is neither pulled from inside the lattice nor replaced before it is read -- the
//     diagonal that points out of the corner (lid-left 7, lid-right 8, bottom-left 6, bottom-right 5: the x rule reads it) -- and
//     MRT_GPU.py's `if` around the push leaves it at the value of the step before (the lattice parks it where the next pull looks
//     for it: update_cell_a, kept slots).  Carried per level next to the density, written back to the parking place at the end.
// Same operations in the same order as update_cell_a / wall_rules / macros on these cells: bit-identical (tests).
// The wall rules of a wall row, applied to the pulled populations in[] of the lane's V cells before the collision (collide_vec then
// takes the row's macroscopic overrides: `kind`).  rw: the cells' previous density (lid; out: this level's); wl / wr: the lane's first / last cell is
// the left / right corner cell of the lattice; kl / kr: that corner's kept slot (in: the value of the step before, out: this step's).
// Only in[] and a few temporaries are live here (outv is not yet): the wall rows cost the kernel no registers.
template <typename R, int V>
__device__ __forceinline__ void wall_row_rules(typename VecT<R, V>::type (&in)[Q], typename VecT<R, V>::type& rw, bool lid, R uLB, bool wl, bool wr,
                                               R& kl, R& kr) {
    typedef typename VecT<R, V>::type T;
    if (lid) {
        // equilibrium of the cell's previous state: rho parked, u = (uLB, 0) -- the expression of equ<>, one direction at a time
        const T ux = T(uLB), uy = T((R)0);
        const T cq = (R)1.5 * (ux * ux + uy * uy);
        auto fek = [&](int k) { const T cu = cu_of<T>(k, ux, uy); return (rw * weight<R>(k)) * ((((R)1. + (R)3.0 * cu) + ((R)4.5 * cu) * cu) - cq); };
        if (wl) {   // x == 0 first (MRT_GPU.py:674-677); slot 7 is the kept one
            const R g7 = kl;
            in[1][0] = (fek(1)[0] - fek(3)[0]) + in[3][0];
            in[5][0] = (fek(5)[0] - fek(7)[0]) + g7;
            in[8][0] = (fek(8)[0] - fek(6)[0]) + in[6][0];
        }
        if (wr) {   // x == X - 1 (MRT_GPU.py:678-682); slot 8 is the kept one
            const R g8 = kr;
            in[3][V - 1] = (-fek(1)[V - 1] + fek(3)[V - 1]) + in[1][V - 1];
            in[6][V - 1] = (-fek(8)[V - 1] + fek(6)[V - 1]) + g8;
            in[7][V - 1] = (-fek(5)[V - 1] + fek(7)[V - 1]) + in[5][V - 1];
        }
        in[4] = (-fek(2) + fek(4)) + in[2];            // y == 0 (MRT_GPU.py:688-692)
        in[7] = (-fek(5) + fek(7)) + in[5];
        in[8] = (-fek(6) + fek(8)) + in[6];
        if (wl) kl = in[7][0];
        if (wr) kr = in[8][V - 1];
        // the cells' density after the lid override (macros; the same expression collide_vec evaluates for kind 1): what the next
        // level's lid rule needs -- computed here, on the two wall rows only, so that the ordinary rows' collision carries nothing for it
        rw = ((in[0] + in[1]) + in[3]) + (R)2. * ((in[2] + in[5]) + in[6]);
    } else {   // bottom wall, at rest: every fe_a - fe_b of the rules is +0
        if (wl) {   // slot 6 is the kept one
            const R g6 = kl;
            in[1][0] = (R)0 + in[3][0];
            in[5][0] = (R)0 + in[7][0];
            in[8][0] = (R)0 + g6;
        }
        if (wr) {   // slot 5 is the kept one
            const R g5 = kr;
            in[3][V - 1] = (R)0 + in[1][V - 1];
            in[6][V - 1] = (R)0 + in[8][V - 1];
            in[7][V - 1] = (R)0 + g5;
        }
        in[2] = (R)0 + in[4];                        // y == Y - 1 (MRT_GPU.py:684-687)
        in[5] = (R)0 + in[7];
        in[6] = (R)0 + in[8];
        if (wl) kl = in[6][0];
        if (wr) kr = in[5][V - 1];
    }
}

// in[] of an update from level l >= 1: own = the row's planes 0, 1, 3; ub = planes 2, 5, 6 of the row below; dn = planes 4, 7, 8 of the
// row above (all at level l).  WALLS: the side-wall rule of an ORDINARY row (MRT_GPU.py:674-682 at rest: f_a = 0 + f_b, as update_vec)
// rides on the lane shifts -- lane 0 of the first strip holds x = 0 and receives 0 + (its own opposite population) in place of what
// would come from beyond the wall, lane 63 of the last strip (x = nx - 1) likewise; in the other strips those lanes are rim.  (A lattice
// narrower than a strip has its right wall in another lane: `narrow`, fixed up explicitly.  Wall ROWS overwrite what they need
// afterwards, wall_row_rules.)
template <typename R, int V, bool WALLS>
__device__ __forceinline__ void build_in(typename VecT<R, V>::type (&in)[Q], typename VecT<R, V>::type o0, typename VecT<R, V>::type o1,
                                         typename VecT<R, V>::type o3, typename VecT<R, V>::type u2, typename VecT<R, V>::type u5,
                                         typename VecT<R, V>::type u6, typename VecT<R, V>::type d4, typename VecT<R, V>::type d7,
                                         typename VecT<R, V>::type d8) {
    in[0] = o0;
    in[2] = u2;
    in[4] = d4;
    if (WALLS) {
        in[1] = shift_from_lower_e<R, V>(o1, (R)0 + o3[1]);            // x = 0:      f1 = 0 + f3,  f3 of the cell = o3[1]
        in[5] = shift_from_lower_e<R, V>(u5, (R)0 + d7[1]);            //             f5 = 0 + f7
        in[8] = shift_from_lower_e<R, V>(d8, (R)0 + u6[1]);            //             f8 = 0 + f6
        in[3] = shift_from_upper_e<R, V>(o3, (R)0 + o1[V - 2]);        // x = nx - 1: f3 = 0 + f1,  f1 of the cell = o1[V - 2]
        in[6] = shift_from_upper_e<R, V>(u6, (R)0 + d8[V - 2]);        //             f6 = 0 + f8
        in[7] = shift_from_upper_e<R, V>(d7, (R)0 + u5[V - 2]);        //             f7 = 0 + f5
    } else {
        in[1] = shift_from_lower<R, V>(o1);
        in[5] = shift_from_lower<R, V>(u5);
        in[8] = shift_from_lower<R, V>(d8);
        in[3] = shift_from_upper<R, V>(o3);
        in[6] = shift_from_upper<R, V>(u6);
        in[7] = shift_from_upper<R, V>(d7);
    }
}

// WALLS: the strip / segment may touch the lattice's walls (k_stream_walls): xs == 0 puts x = 0 in lane 0, the lane whose last
// cell is x = nx - 1 holds the right wall, row 0 / ny - 1 are wall rows; stores go to the lane's cells in [own_lo, own_hi).
// Without WALLS (k_stream: the wall-free interior of a lattice with a frame, the rows between slabs) own_lo / own_hi are unused
// and the stored columns are [xs + R, xs + 64 V - R) below xe.
template <typename R, int COLL, bool TURB, bool WALLS = false, bool SLAB = false>
__device__ __forceinline__ void stream_segment(const R* __restrict__ src, R* __restrict__ dst, const Geo& geo, const Relax<R>& w, R* __restrict__ lds,
                                               int S, int xs, int ya, int yb, int xe, int own_lo = 0, int own_hi = 0) {
    constexpr int V = 16 / (int)sizeof(R), ROW = 64 * V;        // cells of a wave-row
    typedef typename VecT<R, V>::type T;
    // (the wave index is made a scalar: everything derived from it -- the block's row, the row base addresses, the LDS slots -- then
    // lives in SGPRs, and the global accesses take the scalar-base + 32-bit lane offset form)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // blocks are dealt to waves so that consecutive blocks (which update in alternate iterations) sit on the same SIMDs:
    // hardware wave w goes to SIMD w % 4 (cyclically), block slot q = 2 (w % 8) + w / 8 -> each SIMD gets two even and two odd slots
    const int q = ((wv & 7) << 1) | (wv >> 3);
    const int RV = stream_rim(S, V) / V;                          // rim, in vector cells
    // rows that go through the pipeline: the segment's own and S - 1 lead rows on each side (WALLS: none beyond a wall row)
    // (SLAB: its first / last row is a wall row only where the slab holds the lid / the bottom wall; towards a neighbour the lead rows
    // are the neighbour's rows of the deep halo in the ghost rows, as in k_stream's edge segments.  A kernel of its own, k_stream_walls_slab:
    // the lone lattice's sits exactly at the scalar AND the vector register limit, and the two scalars below tip its fast operator into
    // 14 spilled registers -- 424 -> 404 GLUPS)
    const bool wall_lo = WALLS && (!SLAB || geo.y0 == 0), wall_hi = WALLS && (!SLAB || geo.y0 + geo.ny == geo.NY);
    const int y_first = wall_lo ? max(ya - (S - 1), 0) : ya - (S - 1);
    const int y_end = wall_hi ? min(yb + (S - 1), geo.ny) : yb + (S - 1);
    const int y_lid = wall_lo ? 0 : -(1 << 30), y_bot = wall_hi ? geo.ny - 1 : -(1 << 30);   // the wall rows this lattice / slab holds
    const int nb = y_end - y_first;                               // blocks (rows)
    const int x0 = xs + lane * V;
    const bool lane_in = x0 < geo.nx;                             // (the last strip may reach beyond the lattice)
    const bool lane_out = WALLS ? (x0 >= own_lo && x0 < own_hi) : (lane >= RV && lane < 64 - RV && x0 < xe);
    const bool wl = WALLS && x0 == 0, wr = WALLS && x0 + V == geo.nx;   // this lane's first / last cell is a side-wall cell
    const bool side_strip = WALLS && (xs == 0 || xs + ROW >= geo.nx);  // (uniform) the strip holds a side wall
    const bool narrow = WALLS && xs + ROW > geo.nx;                    // (uniform) ... and its right wall is not in lane 63
    R* const slot_mine = lds + (q * 9) * ROW;                     // slots: [0..2] up (k = 2, 5, 6), [3..5] down buffer 0, [6..8] down buffer 1
    R* const up_mine = slot_mine + lane * V;
    const R* const up_below = lds + (((q + 1) & (ST_WAVES - 1)) * 9) * ROW + lane * V;
    const R* const down_above = lds + (((q + ST_WAVES - 1) & (ST_WAVES - 1)) * 9 + 3) * ROW + lane * V;

    T in[Q], outv[Q], hq, hr;
    // address = scalar row base (SGPR pair) + unsigned 32-bit lane offset: the global_load / global_store "saddr" form, one
    // offset VGPR instead of a 64-bit address pair per plane
#ifdef LBM_EXP_NO_HBM
    auto row_base = [&](const R* p, int k, int y) { return (const char*)(p + ((long long)k * geo.plane + (long long)((y & 63) + GHY) * geo.row)); };
#else
    auto row_base = [&](const R* p, int k, int y) { return (const char*)(p + ((long long)k * geo.plane + (long long)(y + GHY) * geo.row)); };
#endif
    const unsigned lane_off = (unsigned)(GH + x0) * (unsigned)sizeof(R);
    auto cell = [&](const char* base, int dx) {
        unsigned off = lane_off;
        asm volatile("" : "+v"(off));      // (opaque: keeps base + offset from being hoisted out of the block loop as a 64-bit per-lane address)
        return (const R*)(base + (off + (unsigned)(dx * (int)sizeof(R))));
    };
    // (WALLS: every register of the prefetch is written on every path -- lanes beyond the lattice load the row's last vector, a block
    // that does not exist gets zeros: a conditionally skipped load keeps the registers' OLD values alive through the whole block before,
    // which at this kernel's register count means spilled, and reloaded behind the new loads, i.e. behind an HBM round trip)
    const unsigned lane_off_ld = WALLS ? (unsigned)(GH + min(x0, max(geo.nx - V, 0))) * (unsigned)sizeof(R) : lane_off;
    auto cell_ld = [&](const char* base, int dx) {
        unsigned off = lane_off_ld;
        asm volatile("" : "+v"(off));
        return (const R*)(base + (off + (unsigned)(dx * (int)sizeof(R))));
    };
    auto load_row = [&](int y, bool exists) {    // the nine pulls of a single step, straight from the lattice (level 0 -> 1)
        if (WALLS) {
            if (exists) {
#pragma unroll
                for (int k = 0; k < Q; ++k) in[k] = vload<R, V, false>(cell_ld(row_base(src, k, y + cyk(k)), -cxk(k)), cxk(k) == 0);
                if (TURB) {
                    hq = vload<R, V, false>(cell_ld(row_base(src, K_QEQ, y), 0), true);
                    hr = vload<R, V, false>(cell_ld(row_base(src, K_RHO, y), 0), true);
                }
            } else {
#pragma unroll
                for (int k = 0; k < Q; ++k) in[k] = T{};
                hq = hr = T{};
            }
            return;
        }
        if (!lane_in || !exists) return;
#pragma unroll
        for (int k = 0; k < Q; ++k) in[k] = vload<R, V, false>(cell(row_base(src, k, y + cyk(k)), -cxk(k)), cxk(k) == 0);
        if (TURB) {
            hq = vload<R, V, false>(cell(row_base(src, K_QEQ, y), 0), true);
            hr = vload<R, V, false>(cell(row_base(src, K_RHO, y), 0), true);
        }
    };
    auto post = [&](int level, bool up, bool down) __attribute__((always_inline)) {    // level reached: 1 .. S - 1
        // (tried in r03 and dropped: the downward planes, which are not read before three iterations later, written at the start of the
        // row's IDLE iteration and fetched by the row below an iteration early, so that half of the LDS traffic leaves the working
        // iterations -- 4096^2 fp32 fast 370.9 -> 370.7 GLUPS, strict 263 -> 270, fp64 fast 174 -> 170, SRT + closure 253 -> 205 (the twelve
        // registers of the early fetch spill there): the LDS phases are not what the working iterations wait for;
        // profiles/r03_logs/lds_rebalance.log)
        R* const dn = up_mine + (3 + (level & 1) * 3) * ROW;
        // (the planes are read into values BEFORE the conditions: with the reads inside them the fp64 build keeps two of the nine
        // output vectors in private memory -- for every row, the ordinary ones included -- and the level loop gains scratch traffic)
        const T p2 = outv[2], p5 = outv[5], p6 = outv[6], p4 = outv[4], p7 = outv[7], p8 = outv[8];
#ifdef LBM_EXP_NO_LDS
        if (geo.nx < 0)
#endif
        if (up) {
            *reinterpret_cast<T*>(up_mine) = p2;
            *reinterpret_cast<T*>(up_mine + ROW) = p5;
            *reinterpret_cast<T*>(up_mine + 2 * ROW) = p6;
        }
#ifdef LBM_EXP_NO_LDS
        if (geo.nx < 0)
#endif
        if (down) {
            *reinterpret_cast<T*>(dn) = p4;
            *reinterpret_cast<T*>(dn + ROW) = p7;
            *reinterpret_cast<T*>(dn + 2 * ROW) = p8;
        }
    };
    // One block: the S updates of row y = y_first + b from the prefetched pulls, the posts, the store.  A wall row (the lid or the
    // bottom wall, WALLS only) keeps the previous density of its cells and the corner cells' kept slot, from one level to the next,
    // in LDS slots of its own that no other row reads: the lid row in its upward post's, the bottom row in its downward post's (buffer
    // 0) -- [0]: densities (a T per lane), [1], [2]: kept slot of the left / right corner cell (an R per lane; only the corner lanes'
    // entries mean anything).  Its wall rules are a short pre-pass on in[] (wall_row_rules); the collision is the same inlined
    // collide_vec as for every row, told the row's kind -- one copy of the arithmetic, no registers on top of the ordinary rows'.
    auto block = [&](int b, auto wrow_tag) __attribute__((always_inline)) {
        constexpr bool WROW = decltype(wrow_tag)::value;   // (a wall row: a copy of the block's code of its own -- nothing of it merges into the ordinary rows' loop)
        const int y = y_first + b;
        const bool wrow = WALLS && WROW, lid = WALLS && WROW && y == 0;
        R* const st = slot_mine + (lid ? 0 : 3 * ROW);
        auto update = [&](bool first) {
            int kind = 0;
            T rw = T{};
            R kl = (R)0, kr = (R)0;
            if (WALLS) {
                if constexpr (WROW) {
                    if (first) {          // from the lattice: the pulls fetched the kept slots from their parking places; the lid cells' parked
                        // densities (wall_rho_at) are loaded here and now -- one wait per launch in the top segments, instead of four
                        // registers carried (and spilled) from every prefetch to every first update
                        if (lid) rw = vload<R, V, false>(cell_ld(row_base(src, 0, -1), 0), true);
                        kl = lid ? in[7][0] : in[6][0];
                        kr = lid ? in[8][V - 1] : in[5][V - 1];
                    } else {
                        if (lid) rw = *reinterpret_cast<const T*>(st + lane * V);
                        kl = st[ROW + lane];
                        kr = st[2 * ROW + lane];
                    }
                    wall_row_rules<R, V>(in, rw, lid, w.uLB, wl, wr, kl, kr);
                    kind = lid ? 1 : 2;
                } else if (side_strip && (first || narrow)) {
                    // side-wall cells of an ordinary row (MRT_GPU.py:674-682 at rest, update_vec): from level 1 on the rule rides on the lane
                    // shifts (build_in); explicitly for the pulls from the lattice, and for the right wall of a lattice narrower than a strip
                    if (wl && first) { in[1][0] = (R)0 + in[3][0]; in[5][0] = (R)0 + in[7][0]; in[8][0] = (R)0 + in[6][0]; }
                    if (wr) { in[3][V - 1] = (R)0 + in[1][V - 1]; in[6][V - 1] = (R)0 + in[8][V - 1]; in[7][V - 1] = (R)0 + in[5][V - 1]; }
                }
            }
#ifdef LBM_EXP_NO_COLLIDE
#pragma unroll
            for (int k = 0; k < Q; ++k) outv[k] = in[k];
#else
            collide_vec<R, COLL, V, TURB>(in, w, outv, hq, hr, wl && kind == 0, wr && kind == 0, kind);
#endif
            if constexpr (WALLS && WROW) {        // (the last level's values are read back from these slots when the row is stored)
                if (lid) *reinterpret_cast<T*>(st + lane * V) = rw;
                st[ROW + lane] = kl;
                st[2 * ROW + lane] = kr;
            }
        };
        const bool up = !lid, down = !(wrow && !lid);                             // (a wall row's unused post: its slots hold the carried wall data)
        update(true);                                                             // level 0 -> 1, from the prefetched pulls
        if (S > 1) post(1, up, down);
        lds_barrier();
        lds_barrier();
        for (int l = 1; l < S; ++l) {                                             // level l -> l + 1
            const R* const dn = down_above + (l & 1) * 3 * ROW;
#ifdef LBM_EXP_NO_LDS
            build_in<R, V, WALLS>(in, outv[0], outv[1], outv[3], outv[2], outv[5], outv[6], outv[4], outv[7], outv[8]);
#else
            build_in<R, V, WALLS>(in, outv[0], outv[1], outv[3], *reinterpret_cast<const T*>(up_below), *reinterpret_cast<const T*>(up_below + ROW),
                                  *reinterpret_cast<const T*>(up_below + 2 * ROW), *reinterpret_cast<const T*>(dn), *reinterpret_cast<const T*>(dn + ROW),
                                  *reinterpret_cast<const T*>(dn + 2 * ROW));
#endif
            update(false);
            if (l + 1 < S) {
                post(l + 1, up, down);
                lds_barrier();
                lds_barrier();
            }
        }
        {   // level S reached: store the row, then -- in the wave's idle iteration -- prefetch its next block
            const T hq_done = hq, hr_done = hr;      // (the prefetch overwrites the history registers)
            T rw_done = T{};
            R kl_done = (R)0, kr_done = (R)0;
            if (WALLS && wrow) {                    // the wall data of the last level, from the row's own LDS slots (update)
                if (lid) rw_done = *reinterpret_cast<const T*>(st + lane * V);
                kl_done = st[ROW + lane];
                kr_done = st[2 * ROW + lane];
            }
            if (lane_out && y >= ya && y < yb) {
#pragma unroll
                for (int k = 0; k < Q; ++k) vstore<R, V, false>(const_cast<R*>(cell(row_base(dst, k, y), 0)), outv[k]);
                if (TURB) {
                    vstore<R, V, false>(const_cast<R*>(cell(row_base(dst, K_QEQ, y), 0)), hq_done);
                    vstore<R, V, false>(const_cast<R*>(cell(row_base(dst, K_RHO, y), 0)), hr_done);
                }
                if (WALLS && wrow) {   // parking places of the lattice format (wall_rho_at; kept slots of update_cell_a)
                    if (lid) {
                        vstore<R, V, false>(const_cast<R*>(cell(row_base(dst, 0, -1), 0)), rw_done);
                        if (wl) *const_cast<R*>(cell(row_base(dst, 7, -1), 1)) = kl_done;
                        if (wr) *const_cast<R*>(cell(row_base(dst, 8, -1), V - 2)) = kr_done;
                    } else {
                        if (wl) *const_cast<R*>(cell(row_base(dst, 6, geo.ny), 1)) = kl_done;
                        if (wr) *const_cast<R*>(cell(row_base(dst, 5, geo.ny), V - 2)) = kr_done;
                    }
                }
            }
            // The prefetch of this wave's next block goes into its IDLE iteration -- behind the barrier that ends the working one: the
            // eighteen address computations and loads no longer lengthen the iteration every other wave waits for (each iteration has one
            // wave that finishes a block), and one iteration still lies between them and the block's first update (r03: 4096^2 fp32 fast
            // 419 -> 445 GLUPS, strict 288 -> 302, fp64 186.5 -> 198 / 129 -> 136; the frame variant k_stream, which had the prefetch AHEAD of
            // the stores: MRT 365 -> 392, TRT 354 -> 368, SRT + closure 259 -> 274; the stores moved to the idle iteration as well: 425 -- they
            // are what the loads queue behind: profiles/r03_logs/whatif.log)
            if (S > 1) lds_barrier();
            load_row(y_first + b + ST_WAVES, b + ST_WAVES < nb);
            if (S > 1) lds_barrier();
        }
        for (int i = 2 * S; i < ST_WAVES; ++i) lds_barrier();
    };
#pragma unroll
    for (int k = 0; k < Q; ++k) in[k] = T{};
    hq = T{}; hr = T{};
    // The schedule of one wave is static: q idle iterations, then per block 16 iterations = S x (update, idle) + 16 - 2 S idle
    // ones, then idle ones up to the common total; every iteration ends with the workgroup barrier (one per iteration for every
    // wave).  Written as loops over blocks and levels -- not as one loop over iterations with a test -- so that the prefetched
    // row (in[]) is live only between two blocks and the carried planes (outv[0], [1], [3]) only inside a block.
    // After its last block a wave simply ends: s_barrier counts the waves of the workgroup that are still alive, so the others go on
    // without it (its posts stay in LDS, which the workgroup keeps to its end).  The workgroup is over nb + 15 iterations after it
    // began -- r02 / early r03 had every wave sit out a common total of ceil(nb / 16) 16 + 16: nothing for a tall segment, 48 -> 37
    // iterations for a slab's edge band, 80 -> 66 for the 37-row segments of a 1024-row lattice.
    for (int i = 0; i < q; ++i) lds_barrier();
    load_row(y_first + q, q < nb);
    for (int b = q; b < nb; b += ST_WAVES) {
        const int y = y_first + b;
        if (WALLS && (SLAB ? (y == y_lid || y == y_bot) : (y == 0 || y == geo.ny - 1))) block(b, std::true_type{});
        else block(b, std::false_type{});
    }
}

// grid: [nframe frame workgroups (a lone lattice)] + nstrips * nsegy segments.  S <= ST_MAX_S steps; F = frame width (>= S, and >= S + 1
// with MRT.py's streaming windows; a multiple of the vector width); a strip's useful columns are [xs + R, xs + 64 V - R), the first strip's start at F.
template <typename R, int COLL, int SEM, bool TURB>
__global__ __launch_bounds__(ST_NT) void k_stream(const R* __restrict__ src, R* __restrict__ dst, Geo geo, Relax<R> w, int S, int F, int xe, int ye,
                                                  int nstrips, int H, FramePtrs<R> fp, int nframe, int nsegx, int nsegy, int seg, int use_lds,
                                                  int lo, int hi, int bands, int xcd_bands) {
    __shared__ __align__(16) R lds[ST_LDS_BYTES / sizeof(R)];
    if ((int)blockIdx.x < nframe) {
        frame_passes<R, COLL, SEM, TURB, ST_NT>(fp, 0, geo, w, F, S, nsegx, nsegy, lo, hi, (int)blockIdx.x, seg, use_lds ? lds : nullptr, bands);
        return;
    }
    constexpr int V = 16 / (int)sizeof(R);
    int b = blockIdx.x - nframe;
    if (xcd_bands) {
        // consecutive workgroups go to consecutive XCDs (each with its own L2): give every XCD a contiguous run of segments, so
        // that the strips that share rim columns -- read at the same time, the workgroups march in step -- share an L2
        const int per = ((int)gridDim.x - nframe) >> 3;
        if (b < (per << 3)) b = (b & 7) * per + (b >> 3);
    }
    const int strip = b % nstrips, sy = b / nstrips;
    const int R_ = stream_rim(S, V);
    const int xs = F - R_ + strip * (64 * V - 2 * R_);
    int ya, yb;
    if (bands) {
        // The edge rows of a slab (the launch that goes before the bulk launch of a unit, see multi_step): the F rows next to an
        // interface are a segment of their own, whose pipeline starts S - 1 rows inside the neighbour's rows of the deep halo
        // (ghost rows); lo / hi = e + 1: e rows of the neighbour's side are owned on top (the lagged lattice, frame_passes).
        const bool top = (bands & 1) && sy == 0;
        if (top) { ya = -(lo > 0 ? lo - 1 : 0); yb = F; }
        else { ya = geo.ny - F; yb = geo.ny + (hi > 0 ? hi - 1 : 0); }
    } else {
        ya = F + sy * H; yb = min(ye, ya + H);
    }
    if (ya >= yb) return;
    stream_segment<R, COLL, TURB>(src, dst, geo, w, lds, S, xs, ya, yb, xe);
}

// The streaming kernel with the walls inside (see "the walls inside the streaming kernel" above): a lone lattice in MRT_GPU.py
// semantics, nx a multiple of the vector width.  grid: nstrips * nsegy segments; strip i starts at min(i TXu, nx - 64 V) with
// TXu = 64 V - 2 R useful columns between two rims (no rim at a wall) and owns the columns [i TXu + R, (i + 1) TXu + R), the
// first one from 0, the last one to nx; segment j owns the rows [j H, (j + 1) H).  No frame, no scratch lattices.
template <typename R, int COLL, bool TURB>
__global__ __launch_bounds__(ST_NT) void k_stream_walls(const R* __restrict__ src, R* __restrict__ dst, Geo geo, Relax<R> w, int S, int nstrips, int H,
                                                        int xcd_bands) {
    __shared__ __align__(16) R lds[ST_LDS_BYTES / sizeof(R)];
    constexpr int V = 16 / (int)sizeof(R), W = 64 * V;
    int b = blockIdx.x;
    if (xcd_bands) {   // (as k_stream: every XCD a contiguous run of segments)
        const int per = (int)gridDim.x >> 3;
        if (b < (per << 3)) b = (b & 7) * per + (b >> 3);
    }
    const int strip = b % nstrips, sy = b / nstrips;
    const int R_ = stream_rim(S, V), TXu = W - 2 * R_;
    const int xs = min(strip * TXu, max(geo.nx - W, 0));
    const int own_lo = strip == 0 ? 0 : strip * TXu + R_;
    const int own_hi = strip == nstrips - 1 ? geo.nx : (strip + 1) * TXu + R_;
    const int ya = sy * H, yb = min(geo.ny, ya + H);
    if (ya >= yb) return;
    stream_segment<R, COLL, TURB, true>(src, dst, geo, w, lds, S, xs, ya, yb, 0, own_lo, own_hi);
}
// ... and for a slab with the deep halo in its ghost rows (r03): the same strips; a unit is two launches (multi_step) -- the edge bands
// (bands != 0: the F rows next to each interface, over the whole width, as segments of their own whose pipeline starts S - 1 rows inside
// the neighbour's rows of the deep halo; lo / hi = e + 1: e rows of the neighbour's side are owned on top -- the lagged lattice; bit 0 =
// the interface above row 0, bit 1 = the one below row ny - 1) and the rows [ybeg, yend) between them, segment j from ybeg + j H.
template <typename R, int COLL, bool TURB>
__global__ __launch_bounds__(ST_NT) void k_stream_walls_slab(const R* __restrict__ src, R* __restrict__ dst, Geo geo, Relax<R> w, int S, int nstrips, int H,
                                                             int xcd_bands, int ybeg, int yend, int bands, int lo, int hi, int F) {
    __shared__ __align__(16) R lds[ST_LDS_BYTES / sizeof(R)];
    constexpr int V = 16 / (int)sizeof(R), W = 64 * V;
    int b = blockIdx.x;
    if (xcd_bands) {
        const int per = (int)gridDim.x >> 3;
        if (b < (per << 3)) b = (b & 7) * per + (b >> 3);
    }
    const int strip = b % nstrips, sy = b / nstrips;
    const int R_ = stream_rim(S, V), TXu = W - 2 * R_;
    const int xs = min(strip * TXu, max(geo.nx - W, 0));
    const int own_lo = strip == 0 ? 0 : strip * TXu + R_;
    const int own_hi = strip == nstrips - 1 ? geo.nx : (strip + 1) * TXu + R_;
    int ya, yb;
    if (bands) {
        const bool top = (bands & 1) && sy == 0;
        if (top) { ya = -(lo > 0 ? lo - 1 : 0); yb = F; }
        else { ya = geo.ny - F; yb = geo.ny + (hi > 0 ? hi - 1 : 0); }
    } else {
        ya = ybeg + sy * H; yb = min(yend, ya + H);
    }
    if (ya >= yb) return;
    stream_segment<R, COLL, TURB, true, true>(src, dst, geo, w, lds, S, xs, ya, yb, 0, own_lo, own_hi);
}
// strips of k_stream_walls for a lattice nx wide (host and device agree through this one function)
__host__ __device__ constexpr int stream_walls_strips(int nx, int S, int V) {
    return nx <= 64 * V ? 1 : (nx - 64 * V + (64 * V - 2 * stream_rim(S, V)) - 1) / (64 * V - 2 * stream_rim(S, V)) + 1;
}

// ---- two rows per wave (k_stream_pairs) -------------------------------------------------------------------------------------------
// k_stream / k_stream_walls hold ONE row per wave; a row updates in every other iteration (its neighbours must catch up in between), so
// half of the sixteen waves wait at the barrier in any iteration -- two issuing waves per SIMD, in lock-step with their LDS traffic --
// and a block always takes sixteen iterations, whatever S (r02: VALU issue 0.47).  Here a wave holds a PAIR of adjacent
// rows, A = 2 j (even, relative to the segment's first row) and B = 2 j + 1.  Row r performs its update number l in iteration
// r + 2 l as before, so A updates in the even and B in the odd iterations of the pair's 2 S: the wave works in every one of them.
//   * the exchange between A and B needs no barrier: A's update takes B's upward planes (2, 5, 6) -- of the level B reached an iteration
//     earlier -- from B's registers; B's update needs A's downward planes (4, 7, 8) of the level BEFORE A's latest update, which A
//     leaves in three LDS slots of the wave's own when it moves on.  A's upward planes (for the pair above) and B's downward planes (for
//     the pair below, two buffers: read three iterations later) are posted as before: 9 reads + 9 writes of LDS per two updates instead
//     of 12 + 12, 12 KiB per wave;
//   * a pair costs 12 VGPRs more than a row (B's resident planes): 120 .. 140, so the workgroup is twelve waves
//     -- three per SIMD at up to 168 VGPRs, all of them issuing -- with S <= W - 2 levels: every wave then has 2 (W - S) >= 4 idle
//     iterations per 2 W, in which it loads its next pair (the 2 x 9 pulls of a single step; nothing else is live then) and parks B's
//     half in LDS slots that are free just then (park_B);
//   * S = 10 steps per launch with W = 12: one read and one write of the lattice per ten steps; W = S + 2 waves for a
//     shorter unit, so a tail of four steps costs about half of an eight-step launch, not the same;
//   * the walls are inside, exactly as in k_stream_walls (same wall_row_rules / collide_vec kinds, same LDS slots for the carried wall
//     data: the lid row is the A row of the first pair, the bottom row the last row of the last pair).
// Same per-cell arithmetic (collide_vec) in the same order: bit-identical to every other kernel.
constexpr int SP_MAX_WAVES = 12;                          // three waves per SIMD
constexpr int SP_NT = SP_MAX_WAVES * 64;

constexpr int SP_MAX_S = SP_MAX_WAVES - 2;                // two idle pair-slots (four iterations) per round to load the next pair in
constexpr int SP_SLOTS = 12;                               // LDS row slots (1 KiB) per wave: 9 posts + 3 for A's downward planes of the level before
constexpr int SP_LDS_BYTES = SP_MAX_WAVES * SP_SLOTS * 1024;   // 144 KiB

template <typename R, int COLL, bool TURB>
__device__ __forceinline__ void stream_pairs_segment(const R* __restrict__ src, R* __restrict__ dst, const Geo& geo, const Relax<R>& w, R* __restrict__ lds,
                                                     int S, int xs, int ya, int yb, int own_lo, int own_hi) {
    constexpr int V = 16 / (int)sizeof(R), ROW = 64 * V;
    typedef typename VecT<R, V>::type T;
    const int W = (int)(blockDim.x >> 6);                         // waves of the workgroup (S + 1 <= W <= SP_MAX_WAVES)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int y_first = max(ya - (S - 1), 0), y_end = min(yb + (S - 1), geo.ny);
    const int nb = y_end - y_first, npairs = (nb + 1) >> 1;       // (an odd row count: the last pair's B row is a dummy that nobody reads)
    const int x0 = xs + lane * V;
    const bool lane_out = x0 >= own_lo && x0 < own_hi;
    const bool wl = x0 == 0, wr = x0 + V == geo.nx;               // this lane's first / last cell is a side-wall cell
    const bool side_strip = xs == 0 || xs + ROW >= geo.nx;        // (uniform) the strip holds a side wall
    const bool narrow = xs + ROW > geo.nx;                        // (uniform) ... and its right wall is not in lane 63
    R* const slot_mine = lds + (wv * SP_SLOTS) * ROW;             // [0..2] A's upward post (k = 2, 5, 6); [3..5], [6..8] B's downward post, two buffers;
    R* const post_mine = slot_mine + lane * V;                    // [9..11] A's downward planes (4, 7, 8) of the level before its latest update (the wave's own)
    const R* const up_below = lds + (((wv + 1 == W ? 0 : wv + 1)) * SP_SLOTS) * ROW + lane * V;       // the A row of the pair below: read by B
    const R* const down_above = lds + (((wv == 0 ? W - 1 : wv - 1)) * SP_SLOTS + 3) * ROW + lane * V; // the B row of the pair above: read by A

    T in[Q], outv[Q], rawB[Q];                                    // in: the row being updated (and A's prefetched pulls); rawB: B's prefetched pulls
    T a0, a1, a3, a4, a7, a8, b0, b1, b3, b2, b5, b6;             // resident planes (see above)
    T hqA, hrA, hqB, hrB;                                         // Smagorinsky history of the two rows
    auto row_base = [&](const R* p, int k, int y) __attribute__((always_inline)) { return (const char*)(p + ((long long)k * geo.plane + (long long)(y + GHY) * geo.row)); };
    const unsigned lane_off = (unsigned)(GH + x0) * (unsigned)sizeof(R);
    auto cell = [&](const char* base, int dx) __attribute__((always_inline)) {
        unsigned off = lane_off;
        asm volatile("" : "+v"(off));
        return (const R*)(base + (off + (unsigned)(dx * (int)sizeof(R))));
    };
    // The 2 x 9 pulls of a single step for the rows of pair j (level 0 -> 1), straight from the lattice -- or zeros when there is no such
    // pair / row.  EVERY register of the prefetch is written on every path (lanes beyond the lattice load the last vector of the row):
    // a conditionally skipped load would keep the registers' old values alive through the whole pair before -- spilled, and reloaded
    // here behind the new loads, i.e. behind an HBM round trip (vmcnt counts in order).
    const unsigned lane_off_ld = (unsigned)(GH + min(x0, max(geo.nx - V, 0))) * (unsigned)sizeof(R);
    auto cell_ld = [&](const char* base, int dx) __attribute__((always_inline)) {
        unsigned off = lane_off_ld;
        asm volatile("" : "+v"(off));
        return (const R*)(base + (off + (unsigned)(dx * (int)sizeof(R))));
    };
    auto load_pair = [&](int j) __attribute__((always_inline)) {
        if (j < npairs) {
            const int yA = y_first + 2 * j, yB = yA + 1;
#pragma unroll
            for (int k = 0; k < Q; ++k) in[k] = vload<R, V, false>(cell_ld(row_base(src, k, yA + cyk(k)), -cxk(k)), cxk(k) == 0);
            if (TURB) {
                hqA = vload<R, V, false>(cell_ld(row_base(src, K_QEQ, yA), 0), true);
                hrA = vload<R, V, false>(cell_ld(row_base(src, K_RHO, yA), 0), true);
            }
            const int yBc = min(yB, y_end - 1);                                     // (no B row: the A row's data again, a dummy nobody reads)
#pragma unroll
            for (int k = 0; k < Q; ++k) rawB[k] = vload<R, V, false>(cell_ld(row_base(src, k, yBc + cyk(k)), -cxk(k)), cxk(k) == 0);
            if (TURB) {
                hqB = vload<R, V, false>(cell_ld(row_base(src, K_QEQ, yBc), 0), true);
                hrB = vload<R, V, false>(cell_ld(row_base(src, K_RHO, yBc), 0), true);
            }
        } else {
#pragma unroll
            for (int k = 0; k < Q; ++k) { in[k] = T{}; rawB[k] = T{}; }
            hqA = hrA = hqB = hrB = T{};
        }
    };
    // B's prefetched pulls wait for their turn in LDS, not in registers: through A's first update they would push the pair over the 168
    // VGPRs (and a spill of just-loaded registers waits for HBM).  The wave's own slots [3..11] are free for it: its downward posts were
    // last read one iteration after the pair before ended, A's old-planes slots are written again at A's second update -- and B's first
    // update, which reads the parked planes, comes before both.  Called two idle iterations after the loads were issued.
    auto park_B = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < Q; ++k) *reinterpret_cast<T*>(post_mine + (3 + k) * ROW) = rawB[k];
    };
    // one update of row y from in[] (level l -> l + 1; first: from the lattice).  WP (compile time): the pair holds a wall row -- such
    // pairs run a copy of the pair's code of their own (pair_body), so that nothing of the wall rows merges into the ordinary pairs' loop
    auto update = [&](int y, bool first, T& hq, T& hr, auto wp_tag) __attribute__((always_inline)) {
        constexpr bool WP = decltype(wp_tag)::value;
        int kind = 0;
        if constexpr (WP) {
            const bool wrow = y == 0 || y == geo.ny - 1, lid = y == 0;
            R* const st = slot_mine + (lid ? 0 : 3 * ROW);
            T rw = T{};
            R kl = (R)0, kr = (R)0;
            if (wrow) {
                if (first) {      // (the lid cells' parked densities, wall_rho_at: loaded here and now, once per launch in the top segments)
                    if (lid) rw = vload<R, V, false>(cell_ld(row_base(src, 0, -1), 0), true);
                    kl = lid ? in[7][0] : in[6][0];
                    kr = lid ? in[8][V - 1] : in[5][V - 1];
                } else {
                    if (lid) rw = *reinterpret_cast<const T*>(st + lane * V);
                    kl = st[ROW + lane];
                    kr = st[2 * ROW + lane];
                }
                wall_row_rules<R, V>(in, rw, lid, w.uLB, wl, wr, kl, kr);
                kind = lid ? 1 : 2;
                if (lid) *reinterpret_cast<T*>(st + lane * V) = rw;
                st[ROW + lane] = kl;
                st[2 * ROW + lane] = kr;
            }
        }
        if (kind == 0 && side_strip && (first || narrow)) {   // (levels >= 1: the side-wall rule rides on the lane shifts, build_in)
            if (wl && first) { in[1][0] = (R)0 + in[3][0]; in[5][0] = (R)0 + in[7][0]; in[8][0] = (R)0 + in[6][0]; }
            if (wr) { in[3][V - 1] = (R)0 + in[1][V - 1]; in[6][V - 1] = (R)0 + in[8][V - 1]; in[7][V - 1] = (R)0 + in[5][V - 1]; }
        }
        collide_vec<R, COLL, V, TURB>(in, w, outv, hq, hr, wl && kind == 0, wr && kind == 0, kind);
    };
    auto store_row = [&](int y, const T& hq, const T& hr) __attribute__((always_inline)) {      // level S reached: the row, and a wall row's parking places
        if (!(lane_out && y >= ya && y < yb)) return;
#pragma unroll
        for (int k = 0; k < Q; ++k) vstore<R, V, false>(const_cast<R*>(cell(row_base(dst, k, y), 0)), outv[k]);
        if (TURB) {
            vstore<R, V, false>(const_cast<R*>(cell(row_base(dst, K_QEQ, y), 0)), hq);
            vstore<R, V, false>(const_cast<R*>(cell(row_base(dst, K_RHO, y), 0)), hr);
        }
        if (y == 0 || y == geo.ny - 1) {
            const bool lid = y == 0;
            const R* const st = slot_mine + (lid ? 0 : 3 * ROW);
            const R kl = st[ROW + lane], kr = st[2 * ROW + lane];
            if (lid) {
                vstore<R, V, false>(const_cast<R*>(cell(row_base(dst, 0, -1), 0)), *reinterpret_cast<const T*>(st + lane * V));
                if (wl) *const_cast<R*>(cell(row_base(dst, 7, -1), 1)) = kl;
                if (wr) *const_cast<R*>(cell(row_base(dst, 8, -1), V - 2)) = kr;
            } else {
                if (wl) *const_cast<R*>(cell(row_base(dst, 6, geo.ny), 1)) = kl;
                if (wr) *const_cast<R*>(cell(row_base(dst, 5, geo.ny), V - 2)) = kr;
            }
        }
    };
#pragma unroll
    for (int k = 0; k < Q; ++k) { in[k] = T{}; rawB[k] = T{}; }
    hqA = hrA = hqB = hrB = T{};
    a0 = a1 = a3 = a4 = a7 = a8 = b0 = b1 = b3 = b2 = b5 = b6 = T{};
    // The schedule of a wave: 2 wv idle iterations, then per pair 2 S working iterations (A, B, A, B, ...) and 2 (W - S) idle ones
    // (the next pair is loaded at their start), then idle ones up to the common total; one workgroup barrier per iteration.
    const int rounds = (npairs + W - 1) / W;
    const int total = 2 * W * rounds + 2 * W;
    int done = 2 * wv;
    load_pair(wv);
    for (int i = 0; i < 2 * wv; ++i) {
        if (i == 2) park_B();
        lds_barrier();
    }
    if (2 * wv <= 2) park_B();          // (the first waves start at once: they wait for their loads here, once)
    auto pair_body = [&](int j, auto wp_tag) __attribute__((always_inline)) {
        const int yA = y_first + 2 * j, yB = yA + 1;
        const bool lidA = yA == 0, botB = yB >= geo.ny - 1;       // (their unused posts' slots hold the carried wall data -- also when the
                                                                  //  bottom row is the A row and B a dummy beyond the lattice)
        // what follows an update of A / B that reached level l1 = l + 1
        auto after_A = [&](int l1) __attribute__((always_inline)) {
            a0 = outv[0]; a1 = outv[1]; a3 = outv[3]; a4 = outv[4]; a7 = outv[7]; a8 = outv[8];
            const T p2 = outv[2], p5 = outv[5], p6 = outv[6];     // (read before the conditions: see stream_segment's post)
            if (l1 < S) {
                if (!lidA) {
                    *reinterpret_cast<T*>(post_mine) = p2;
                    *reinterpret_cast<T*>(post_mine + ROW) = p5;
                    *reinterpret_cast<T*>(post_mine + 2 * ROW) = p6;
                }
            } else {
                store_row(yA, hqA, hrA);
            }
            lds_barrier();
        };
        auto after_B = [&](int l1) __attribute__((always_inline)) {
            b0 = outv[0]; b1 = outv[1]; b3 = outv[3]; b2 = outv[2]; b5 = outv[5]; b6 = outv[6];
            const T p4 = outv[4], p7 = outv[7], p8 = outv[8];
            if (l1 < S) {
                if (!botB) {
                    R* const dn = post_mine + (3 + (l1 & 1) * 3) * ROW;
                    *reinterpret_cast<T*>(dn) = p4;
                    *reinterpret_cast<T*>(dn + ROW) = p7;
                    *reinterpret_cast<T*>(dn + 2 * ROW) = p8;
                }
            }
            lds_barrier();
        };
        // level 0 -> 1 of both rows, from the prefetched pulls (outside the loop: rawB must not stay live through the levels)
        update(yA, true, hqA, hrA, wp_tag);
        after_A(1);
#pragma unroll
        for (int k = 0; k < Q; ++k) in[k] = *reinterpret_cast<const T*>(post_mine + (3 + k) * ROW);    // (parked by park_B)
        update(yB, true, hqB, hrB, wp_tag);
        after_B(1);
        for (int l = 1; l < S; ++l) {
            {   // ---- A: level l -> l + 1
                const R* const dn = down_above + (l & 1) * 3 * ROW;
                build_in<R, V, true>(in, a0, a1, a3, b2, b5, b6, *reinterpret_cast<const T*>(dn), *reinterpret_cast<const T*>(dn + ROW),
                                     *reinterpret_cast<const T*>(dn + 2 * ROW));
                // A's downward planes of level l: B's update of the next iteration reads them -- through three LDS slots of the wave's own
                // (no barrier involved) rather than twelve more registers: with the packed constants of the operator resident the pair
                // would not fit the 168 VGPRs of three waves per SIMD, and a spill reloaded behind the next pair's prefetch waits for HBM
                *reinterpret_cast<T*>(post_mine + 9 * ROW) = a4;
                *reinterpret_cast<T*>(post_mine + 10 * ROW) = a7;
                *reinterpret_cast<T*>(post_mine + 11 * ROW) = a8;
                update(yA, false, hqA, hrA, wp_tag);
                after_A(l + 1);
            }
            {   // ---- B: level l -> l + 1
                build_in<R, V, true>(in, b0, b1, b3, *reinterpret_cast<const T*>(up_below), *reinterpret_cast<const T*>(up_below + ROW),
                                     *reinterpret_cast<const T*>(up_below + 2 * ROW), *reinterpret_cast<const T*>(post_mine + 9 * ROW),
                                     *reinterpret_cast<const T*>(post_mine + 10 * ROW), *reinterpret_cast<const T*>(post_mine + 11 * ROW));
                update(yB, false, hqB, hrB, wp_tag);
                after_B(l + 1);
            }
        }
        {   // the pair is done (outv: B's row at level S): load this wave's next pair -- OUTSIDE the level loop, or the 72 registers it
            // fills would be live through every level -- then store B (A's row was stored when it reached level S)
            const T hq_done = hqB, hr_done = hrB;
            load_pair(j + W);
            store_row(yB, hq_done, hr_done);
        }
    };
    for (int j = wv; j < npairs; j += W) {
        const int yA = y_first + 2 * j;
        if (yA == 0 || yA + 1 >= geo.ny - 1) pair_body(j, std::true_type{});      // (the lid row, or the bottom row as A or B)
        else pair_body(j, std::false_type{});
        for (int i = 2 * S; i < 2 * W; ++i) {
            if (i == 2 * S + 2) park_B();       // (the next pair's B row, two iterations after its loads were issued)
            lds_barrier();
        }
        done += 2 * W;
    }
    for (; done < total; ++done) lds_barrier();
}

// grid: nstrips * nsegy segments (strips and owned columns as k_stream_walls); block: 64 W threads, S + 1 <= W <= 12.
template <typename R, int COLL, bool TURB>
__global__ __launch_bounds__(SP_NT) void k_stream_pairs(const R* __restrict__ src, R* __restrict__ dst, Geo geo, Relax<R> w, int S, int nstrips, int H,
                                                        int xcd_bands) {
    __shared__ __align__(16) R lds[SP_LDS_BYTES / sizeof(R)];
    constexpr int V = 16 / (int)sizeof(R), W = 64 * V;
    int b = blockIdx.x;
    if (xcd_bands) {
        const int per = (int)gridDim.x >> 3;
        if (b < (per << 3)) b = (b & 7) * per + (b >> 3);
    }
    const int strip = b % nstrips, sy = b / nstrips;
    const int R_ = stream_rim(S, V), TXu = W - 2 * R_;
    const int xs = min(strip * TXu, max(geo.nx - W, 0));
    const int own_lo = strip == 0 ? 0 : strip * TXu + R_;
    const int own_hi = strip == nstrips - 1 ? geo.nx : (strip + 1) * TXu + R_;
    const int ya = sy * H, yb = min(geo.ny, ya + H);
    if (ya >= yb) return;
    stream_pairs_segment<R, COLL, TURB>(src, dst, geo, w, lds, S, xs, ya, yb, own_lo, own_hi);
}

#ifndef LBM_STREAMP_EXTERN
#define LBM_STREAMP_EXTERN extern
#endif
#define LBM_STREAMP_ONE(R, COLL, TURB)                                                                                    \
    LBM_STREAMP_EXTERN template __global__ void k_stream_pairs<R, COLL, TURB>(const R* __restrict__, R* __restrict__, Geo, Relax<R>, int, int, int, int);
#define LBM_STREAMP_ALL(R)                                                                                                \
    LBM_STREAMP_ONE(R, C_SRT, false) LBM_STREAMP_ONE(R, C_TRT, false) LBM_STREAMP_ONE(R, C_MRT, false)                      \
    LBM_STREAMP_ONE(R, C_MRT_FAST, false) LBM_STREAMP_ONE(R, C_SRT_FAST, false) LBM_STREAMP_ONE(R, C_TRT_FAST, false)       \
    LBM_STREAMP_ONE(R, C_SRT, true) LBM_STREAMP_ONE(R, C_TRT, true) LBM_STREAMP_ONE(R, C_MRT, true)                         \
    LBM_STREAMP_ONE(R, C_MRT_FAST, true) LBM_STREAMP_ONE(R, C_SRT_FAST, true) LBM_STREAMP_ONE(R, C_TRT_FAST, true)
#if !defined(LBM_STREAM_ONLY_F64) && !defined(LBM_STREAMP_SKIP)
LBM_STREAMP_ALL(float)
#endif
#if !defined(LBM_STREAM_ONLY_F32) && !defined(LBM_STREAMP_SKIP)
LBM_STREAMP_ALL(double)
#endif

// explicit instantiations live in lbm_stream_f32.hip / lbm_stream_f64.hip and lbm_streamw_f32.hip / lbm_streamw_f64.hip
// (LBM_STREAM_EXTERN / LBM_STREAMW_EXTERN empty there)
#ifndef LBM_STREAMW_EXTERN
#define LBM_STREAMW_EXTERN extern
#endif
#define LBM_STREAMW_ONE(R, COLL, TURB)                                                                                    \
    LBM_STREAMW_EXTERN template __global__ void k_stream_walls<R, COLL, TURB>(const R* __restrict__, R* __restrict__, Geo, Relax<R>, int, int, int, int);
#define LBM_STREAMW_ALL(R)                                                                                                \
    LBM_STREAMW_ONE(R, C_SRT, false) LBM_STREAMW_ONE(R, C_TRT, false) LBM_STREAMW_ONE(R, C_MRT, false)                      \
    LBM_STREAMW_ONE(R, C_MRT_FAST, false) LBM_STREAMW_ONE(R, C_SRT_FAST, false) LBM_STREAMW_ONE(R, C_TRT_FAST, false)       \
    LBM_STREAMW_ONE(R, C_SRT, true) LBM_STREAMW_ONE(R, C_TRT, true) LBM_STREAMW_ONE(R, C_MRT, true)                         \
    LBM_STREAMW_ONE(R, C_MRT_FAST, true) LBM_STREAMW_ONE(R, C_SRT_FAST, true) LBM_STREAMW_ONE(R, C_TRT_FAST, true)
#if !defined(LBM_STREAM_ONLY_F64) && !defined(LBM_STREAMW_SKIP)
LBM_STREAMW_ALL(float)
#endif
#if !defined(LBM_STREAM_ONLY_F32) && !defined(LBM_STREAMW_SKIP)
LBM_STREAMW_ALL(double)
#endif

// ... the slab variant of the same: lbm_streams_f32.hip / lbm_streams_f64.hip
#ifndef LBM_STREAMS_EXTERN
#define LBM_STREAMS_EXTERN extern
#endif
#define LBM_STREAMS_ONE(R, COLL, TURB)                                                                                    \
    LBM_STREAMS_EXTERN template __global__ void k_stream_walls_slab<R, COLL, TURB>(const R* __restrict__, R* __restrict__, Geo, Relax<R>, int, int, int, \
                                                                                   int, int, int, int, int, int, int);
#define LBM_STREAMS_ALL(R)                                                                                                \
    LBM_STREAMS_ONE(R, C_SRT, false) LBM_STREAMS_ONE(R, C_TRT, false) LBM_STREAMS_ONE(R, C_MRT, false)                      \
    LBM_STREAMS_ONE(R, C_MRT_FAST, false) LBM_STREAMS_ONE(R, C_SRT_FAST, false) LBM_STREAMS_ONE(R, C_TRT_FAST, false)       \
    LBM_STREAMS_ONE(R, C_SRT, true) LBM_STREAMS_ONE(R, C_TRT, true) LBM_STREAMS_ONE(R, C_MRT, true)                         \
    LBM_STREAMS_ONE(R, C_MRT_FAST, true) LBM_STREAMS_ONE(R, C_SRT_FAST, true) LBM_STREAMS_ONE(R, C_TRT_FAST, true)
#if !defined(LBM_STREAM_ONLY_F64) && !defined(LBM_STREAMS_SKIP)
LBM_STREAMS_ALL(float)
#endif
#if !defined(LBM_STREAM_ONLY_F32) && !defined(LBM_STREAMS_SKIP)
LBM_STREAMS_ALL(double)
#endif

#ifndef LBM_STREAM_EXTERN
#define LBM_STREAM_EXTERN extern
#endif
#define LBM_STREAM_ONE(R, COLL, SEM, TURB)                                                                               \
    LBM_STREAM_EXTERN template __global__ void k_stream<R, COLL, SEM, TURB>(const R* __restrict__, R* __restrict__, Geo, Relax<R>, int, int, int, int, \
                                                                          int, int, FramePtrs<R>, int, int, int, int, int, int, int, int, int);
#define LBM_STREAM_ALL(R)                                                                                                \
    LBM_STREAM_ONE(R, C_SRT, SEM_GPU, false) LBM_STREAM_ONE(R, C_TRT, SEM_GPU, false) LBM_STREAM_ONE(R, C_MRT, SEM_GPU, false)          \
    LBM_STREAM_ONE(R, C_MRT_FAST, SEM_GPU, false) LBM_STREAM_ONE(R, C_SRT_FAST, SEM_GPU, false) LBM_STREAM_ONE(R, C_TRT_FAST, SEM_GPU, false) \
    LBM_STREAM_ONE(R, C_SRT, SEM_GPU, true) LBM_STREAM_ONE(R, C_TRT, SEM_GPU, true) LBM_STREAM_ONE(R, C_MRT, SEM_GPU, true)             \
    LBM_STREAM_ONE(R, C_MRT_FAST, SEM_GPU, true) LBM_STREAM_ONE(R, C_SRT_FAST, SEM_GPU, true) LBM_STREAM_ONE(R, C_TRT_FAST, SEM_GPU, true) \
    LBM_STREAM_ONE(R, C_SRT, SEM_PY, false) LBM_STREAM_ONE(R, C_TRT, SEM_PY, false) LBM_STREAM_ONE(R, C_MRT, SEM_PY, false)
#if !defined(LBM_STREAM_ONLY_F64) && !defined(LBM_STREAM_SKIP)
LBM_STREAM_ALL(float)
#endif
#if !defined(LBM_STREAM_ONLY_F32) && !defined(LBM_STREAM_SKIP)
LBM_STREAM_ALL(double)
#endif
