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
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
// takes the row's macroscopic overrides: `kind`).  rw: the cells' previous density (lid); wl / wr: the lane's first / last cell is
// the left / right corner cell of the lattice; kl / kr: that corner's kept slot (in: the value of the step before, out: this step's).
// Only in[] and a few temporaries are live here (outv is not yet): the wall rows cost the kernel no registers.
template <typename R, int V>
__device__ __forceinline__ void wall_row_rules(typename VecT<R, V>::type (&in)[Q], typename VecT<R, V>::type rw, bool lid, R uLB, bool wl, bool wr,
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

// WALLS: the strip / segment may touch the lattice's walls (k_stream_walls): xs == 0 puts x = 0 in lane 0, the lane whose last
// cell is x = nx - 1 holds the right wall, row 0 / ny - 1 are wall rows; stores go to the lane's cells in [own_lo, own_hi).
// Without WALLS (k_stream: the wall-free interior of a lattice with a frame, the rows between slabs) own_lo / own_hi are unused
// and the stored columns are [xs + R, xs + 64 V - R) below xe.
template <typename R, int COLL, bool TURB, bool WALLS = false>
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
    const int y_first = WALLS ? max(ya - (S - 1), 0) : ya - (S - 1);
    const int y_end = WALLS ? min(yb + (S - 1), geo.ny) : yb + (S - 1);
    const int nb = y_end - y_first;                               // blocks (rows)
    const int x0 = xs + lane * V;
    const bool lane_in = x0 < geo.nx;                             // (the last strip may reach beyond the lattice)
    const bool lane_out = WALLS ? (x0 >= own_lo && x0 < own_hi) : (lane >= RV && lane < 64 - RV && x0 < xe);
    const bool wl = WALLS && x0 == 0, wr = WALLS && x0 + V == geo.nx;   // this lane's first / last cell is a side-wall cell
    R* const slot_mine = lds + (q * 9) * ROW;                     // slots: [0..2] up (k = 2, 5, 6), [3..5] down buffer 0, [6..8] down buffer 1
    R* const up_mine = slot_mine + lane * V;
    const R* const up_below = lds + (((q + 1) & (ST_WAVES - 1)) * 9) * ROW + lane * V;
    const R* const down_above = lds + (((q + ST_WAVES - 1) & (ST_WAVES - 1)) * 9 + 3) * ROW + lane * V;

    T in[Q], outv[Q], hq, hr, rwp;
    // address = scalar row base (SGPR pair) + unsigned 32-bit lane offset: the global_load / global_store "saddr" form, one
    // offset VGPR instead of a 64-bit address pair per plane
    auto row_base = [&](const R* p, int k, int y) { return (const char*)(p + ((long long)k * geo.plane + (long long)(y + GHY) * geo.row)); };
    const unsigned lane_off = (unsigned)(GH + x0) * (unsigned)sizeof(R);
    auto cell = [&](const char* base, int dx) {
        unsigned off = lane_off;
        asm volatile("" : "+v"(off));      // (opaque: keeps base + offset from being hoisted out of the block loop as a 64-bit per-lane address)
        return (const R*)(base + (off + (unsigned)(dx * (int)sizeof(R))));
    };
    auto load_row = [&](int y) {    // the nine pulls of a single step, straight from the lattice (level 0 -> 1)
        if (!lane_in) return;
#pragma unroll
        for (int k = 0; k < Q; ++k) in[k] = vload<R, V, false>(cell(row_base(src, k, y + cyk(k)), -cxk(k)), cxk(k) == 0);
        if (TURB) {
            hq = vload<R, V, false>(cell(row_base(src, K_QEQ, y), 0), true);
            hr = vload<R, V, false>(cell(row_base(src, K_RHO, y), 0), true);
        }
        if (WALLS && y == 0) rwp = vload<R, V, false>(cell(row_base(src, 0, -1), 0), true);   // the lid cells' parked densities (wall_rho_at)
    };
    auto post = [&](int level, bool up, bool down) {    // level reached: 1 .. S - 1
        R* const dn = up_mine + (3 + (level & 1) * 3) * ROW;
        if (up) {
            *reinterpret_cast<T*>(up_mine) = outv[2];
            *reinterpret_cast<T*>(up_mine + ROW) = outv[5];
            *reinterpret_cast<T*>(up_mine + 2 * ROW) = outv[6];
        }
        if (down) {
            *reinterpret_cast<T*>(dn) = outv[4];
            *reinterpret_cast<T*>(dn + ROW) = outv[7];
            *reinterpret_cast<T*>(dn + 2 * ROW) = outv[8];
        }
    };
    // One block: the S updates of row y = y_first + b from the prefetched pulls, the posts, the store.  A wall row (the lid or the
    // bottom wall, WALLS only) keeps the previous density of its cells and the corner cells' kept slot, from one level to the next,
    // in LDS slots of its own that no other row reads: the lid row in its upward post's, the bottom row in its downward post's (buffer
    // 0) -- [0]: densities (a T per lane), [1], [2]: kept slot of the left / right corner cell (an R per lane; only the corner lanes'
    // entries mean anything).  Its wall rules are a short pre-pass on in[] (wall_row_rules); the collision is the same inlined
    // collide_vec as for every row, told the row's kind -- one copy of the arithmetic, no registers on top of the ordinary rows'.
    auto block = [&](int b) {
        const int y = y_first + b;
        const bool wrow = WALLS && (y == 0 || y == geo.ny - 1), lid = WALLS && y == 0;
        R* const st = slot_mine + (lid ? 0 : 3 * ROW);
        auto update = [&](bool first) {
            int kind = 0;
            T rw = T{};
            R kl = (R)0, kr = (R)0;
            if (WALLS) {
                if (wrow) {
                    if (first) {          // from the lattice: the pulls fetched the kept slots from their parking places
                        rw = rwp;
                        kl = lid ? in[7][0] : in[6][0];
                        kr = lid ? in[8][V - 1] : in[5][V - 1];
                    } else {
                        if (lid) rw = *reinterpret_cast<const T*>(st + lane * V);
                        kl = st[ROW + lane];
                        kr = st[2 * ROW + lane];
                    }
                    wall_row_rules<R, V>(in, rw, lid, w.uLB, wl, wr, kl, kr);
                    kind = lid ? 1 : 2;
                } else {   // side-wall cells of an ordinary row: MRT_GPU.py:674-682 at rest (update_vec)
                    if (wl) { in[1][0] = (R)0 + in[3][0]; in[5][0] = (R)0 + in[7][0]; in[8][0] = (R)0 + in[6][0]; }
                    if (wr) { in[3][V - 1] = (R)0 + in[1][V - 1]; in[6][V - 1] = (R)0 + in[8][V - 1]; in[7][V - 1] = (R)0 + in[5][V - 1]; }
                }
            }
            collide_vec<R, COLL, V, TURB>(in, w, outv, hq, hr, wl && kind == 0, wr && kind == 0, kind, WALLS ? &rw : nullptr);
            if (WALLS && wrow) {        // (the last level's values are read back from these slots when the row is stored)
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
            in[0] = outv[0];
            in[1] = shift_from_lower<R, V>(outv[1]);
            in[3] = shift_from_upper<R, V>(outv[3]);
            in[2] = *reinterpret_cast<const T*>(up_below);
            in[5] = shift_from_lower<R, V>(*reinterpret_cast<const T*>(up_below + ROW));
            in[6] = shift_from_upper<R, V>(*reinterpret_cast<const T*>(up_below + 2 * ROW));
            in[4] = *reinterpret_cast<const T*>(dn);
            in[7] = shift_from_upper<R, V>(*reinterpret_cast<const T*>(dn + ROW));
            in[8] = shift_from_lower<R, V>(*reinterpret_cast<const T*>(dn + 2 * ROW));
            update(false);
            if (l + 1 < S) {
                post(l + 1, up, down);
                lds_barrier();
                lds_barrier();
            }
        }
        {   // level S reached: prefetch this wave's next block (first: nothing it waits for is behind the stores), then store
            const T hq_done = hq, hr_done = hr;      // (the prefetch overwrites the history registers)
            T rw_done = T{};
            R kl_done = (R)0, kr_done = (R)0;
            if (WALLS && wrow) {                    // the wall data of the last level, from the row's own LDS slots (update)
                if (lid) rw_done = *reinterpret_cast<const T*>(st + lane * V);
                kl_done = st[ROW + lane];
                kr_done = st[2 * ROW + lane];
            }
            if (b + ST_WAVES < nb) load_row(y_first + b + ST_WAVES);
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
            if (S > 1) { lds_barrier(); lds_barrier(); }
        }
        for (int i = 2 * S; i < ST_WAVES; ++i) lds_barrier();
    };
#pragma unroll
    for (int k = 0; k < Q; ++k) in[k] = T{};
    hq = T{}; hr = T{}; rwp = T{};
    // The schedule of one wave is static: q idle iterations, then per block 16 iterations = S x (update, idle) + 16 - 2 S idle
    // ones, then idle ones up to the common total; every iteration ends with the workgroup barrier (one per iteration for every
    // wave).  Written as loops over blocks and levels -- not as one loop over iterations with a test -- so that the prefetched
    // row (in[]) is live only between two blocks and the carried planes (outv[0], [1], [3]) only inside a block.
    const int jtot = (nb + ST_WAVES - 1) / ST_WAVES * ST_WAVES + ST_WAVES;
    int done = q;
    for (int i = 0; i < q; ++i) lds_barrier();
    if (q < nb) load_row(y_first + q);
    for (int b = q; b < nb; b += ST_WAVES) {
        block(b);
        done += ST_WAVES;
    }
    for (; done < jtot; ++done) lds_barrier();
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
// strips of k_stream_walls for a lattice nx wide (host and device agree through this one function)
__host__ __device__ constexpr int stream_walls_strips(int nx, int S, int V) {
    return nx <= 64 * V ? 1 : (nx - 64 * V + (64 * V - 2 * stream_rim(S, V)) - 1) / (64 * V - 2 * stream_rim(S, V)) + 1;
}

// explicit instantiations live in lbm_stream_f32.hip / lbm_stream_f64.hip and lbm_streamw_f32.hip / lbm_streamw_f64.hip
// (LBM_STREAM_EXTERN / LBM_STREAMW_EXTERN empty there)
#ifndef LBM_SINGLE_TU
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
#endif

#ifndef LBM_SINGLE_TU
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
#endif  // LBM_SINGLE_TU
