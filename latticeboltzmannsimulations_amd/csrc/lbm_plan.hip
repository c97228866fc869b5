// lbm_plan.hip -- parameter validation, the launch plan of a context (kernel, steps per launch, strips, workgroup order), the unit
// sequence of a call, and the device-free dry run of all of it (lbm_plan) that bench.py --gpus N compares across ranks.
#include "lbm_host.hpp"

namespace lbmhost {

// All S frame passes lat[from] -> lat[to] in one launch (k_frame_multi); lo / hi: the slab has a neighbour below row 0 / above
// row ny - 1 whose rows lie in the ghost rows (deep halo).
// Do the LDS windows of the fused frame passes fit (two buffers of the largest pass-1 rectangle plus its ring)?
bool frame_lds_fits(const lbm_ctx* c, int S, bool deep_rows, int extra, long long budget) {
    if (!c->frame_lds) return false;
    const int F = c->tb_f, L = c->frame_seg, m = S - 1, np = c->p.turb ? Q + 2 : Q;
    const long long row_strip = (long long)(L + 2 * m + 2) * (F + m + (deep_rows ? m + extra : 0) + 2);
    const long long col_strip = (long long)(F + m + 2) * (L + 2 * m + extra + 2);
    return 2 * np * std::max(row_strip, col_strip) * c->es <= budget;
}
// waves per workgroup of k_stream_pairs for S steps per launch: two idle pair-slots to load the next pair in
int pairs_waves(int S) { return std::min(SP_MAX_WAVES, S + 2); }
StreamPlan plan_stream_on(const lbm_ctx* c, int S, int ncu, long long* cost_out) {
    const int V = 16 / c->es, Rr = stream_rim(S, V), TXu = 64 * V - 2 * Rr, F = c->stream_walls ? 0 : c->tb_f;
    const int cols = c->geo.nx - 2 * F;
    // (with the walls inside: strips over the whole width, no rim at a wall; segments over the whole height -- of a slab: over the rows
    // between its edge bands, tb_f rows next to each interface)
    const int rows = c->stream_walls ? c->geo.ny - c->tb_f * ((has_neighbour(c, LBM_SIDE_LOW) ? 1 : 0) + (has_neighbour(c, LBM_SIDE_HIGH) ? 1 : 0))
                                     : c->geo.ny - 2 * F;
    StreamPlan best{c->stream_walls ? stream_walls_strips(c->geo.nx, S, V) : (cols + TXu - 1) / TXu, 1, rows};
    long long best_cost = -1;
    for (int n = 1; n <= 256 && n * 8 <= std::max(rows, 8); ++n) {
        const int H = (rows + n - 1) / n, nseg = (rows + H - 1) / H;
        const long long segs = (long long)best.nstrips * nseg, rounds = (segs + ncu - 1) / ncu;
        long long iters = H + 2 * (S - 1) + ST_WAVES - 1;      // (block b starts in iteration b and takes 16: stream_segment)
        if (c->stream_pairs) {   // W waves, a pair of rows each: 2 W iterations per W pairs
            const long long Wv = pairs_waves(S), np = (H + 2 * (S - 1) + 1) / 2;
            iters = 2 * Wv * ((np + Wv - 1) / Wv) + 2 * Wv;
        }
        const long long cost = rounds * iters;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best.nsegy = nseg; best.H = H; }
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}
// Between slabs the unit has an edge launch beside the bulk launch (multi_step).  A bulk launch of ONE round takes every CU for its
// whole run, the edge workgroups (a CU each, ~48 pipeline iterations) then run after it and the next exchange after them: the
// unit costs bulk + edges.  For a short slab it is cheaper to plan the bulk launch on fewer CUs and leave the others to the edge
// workgroups (4096 x 512 fp32 slab in loopback: 145 -> 176 GLUPS; taller slabs lose a few per cent -- 4096 x 1024 247 -> 240, 4096 x
// 2048 282 -> 270 -- hence the limit below; profiles/r02_logs/slab_loopback9.log).  Costs in pipeline iterations.
StreamPlan plan_stream(const lbm_ctx* c, int S) {
    long long cost0 = 0;
    const StreamPlan p0 = plan_stream_on(c, S, c->ncu, &cost0);
    if (!(is_slab(c) && c->deep_halo && (c->frame_fused || c->stream_walls) && c->edge_reserve) || S < 3) return p0;
    if ((long long)p0.nstrips * p0.nsegy > c->ncu) return p0;     // several rounds: the bulk launch is released behind the edge launch instead
    const int nb = (has_neighbour(c, LBM_SIDE_LOW) ? 1 : 0) + (has_neighbour(c, LBM_SIDE_HIGH) ? 1 : 0), L = c->frame_seg;
    const long long n_edge = c->stream_walls ? (long long)nb * p0.nstrips     // (the walls inside: the interface bands alone)
                                             : (long long)nb * p0.nstrips + 2LL * ((c->geo.ny + L - 1) / L) + (2LL - nb) * ((c->geo.nx + L - 1) / L);
    const long long edge_it = c->tb_f + 2 * (S - 1) + ST_WAVES - 1;
    // (measured: a bulk launch longer than ~1.5 edge workgroups overlaps the frame variant's ~100 short edge workgroups well enough as
    // it is.  With the walls inside the edge launch is the 2 x nstrips band workgroups alone, each of which holds a CU -- all its LDS --
    // for edge_it iterations, and the bulk workgroups that find no CU start that much later: leaving them room pays up to a bulk launch
    // of ~3 edge workgroups -- 4096 x 512 fp32 slab in loopback 141 -> 186 GLUPS, 4096 x 1024 210 -> 245, 4096 x 2048 289 -> 274:
    // profiles/r03_logs/slab_walls.log)
    if (c->stream_walls ? cost0 > 3 * edge_it : 4 * cost0 > 7 * edge_it) return p0;   // (the frame variant: segments of up to ~35 rows, as measured in r02)
    StreamPlan best = p0;
    long long best_cost = cost0 + edge_it * ((n_edge + c->ncu - 1) / c->ncu);
    // (workgroups go to the eight XCDs in turn and, inside an XCD, to its four shader engines in turn, whatever is free where: "room" is
    // per shader engine, 8 CUs.  Kernel trace of the 4096 x 1024 slab, 34 + 221 workgroups on 256 CUs: XCD 0 is dealt 5 + 28 and its last
    // bulk workgroup starts when an edge workgroup ends, 43 us late; 34 + 204 -- 31 per XCD at most -- still waits, 34 + 187 does not
    // (a sweep of the CUs left free: 40, 48 -> 209 GLUPS, 56, 64 -> 236, 72 -> 222: profiles/r03_logs/slab_walls.log).  So the CUs left
    // to the edge workgroups are counted in units of 32, rounded up.)
    const long long grain = c->stream_walls ? 32 : 1;
    auto in_grains = [&](long long n) { return (n + grain - 1) / grain * grain; };
    if (c->stream_walls && in_grains((long long)p0.nstrips * p0.nsegy) + in_grains(n_edge) <= c->ncu) best_cost = std::max(cost0, edge_it);   // (room for both)
    // (div > 1: the edge workgroups in several rounds on fewer CUs -- the frame variant's many short ones; the band workgroups of the
    // walls variant are all dispatched at once, ahead of the bulk launch, and hold what they get)
    for (int div = 1; div <= (c->stream_walls ? 1 : 3); ++div) {
        const long long r = in_grains((n_edge + div - 1) / div);
        if (r < 1 || r > c->ncu / 2) continue;
        long long cb = 0;
        const StreamPlan p = plan_stream_on(c, S, c->ncu - (int)r, &cb);
        const long long cost = std::max(cb, edge_it * ((n_edge + r - 1) / r));
        if (cost < best_cost) { best_cost = cost; best = p; }
    }
    return best;
}

// Can the lattice of the step before the last be recomputed after a unit of S steps (lazy lag)?  A lone lattice: always (any
// number of single or multi-step launches).  A slab: from the deep halo the unit received, with one launch of S - 1 >= 3 steps.
bool lag_replayable(const lbm_ctx* c, int S) {
    if (S <= 1) return true;
    if (!c->lazy_lag || c->push) return false;
    if (!is_slab(c)) return true;
    return c->deep_halo && S - 1 >= 3;
}

// Steps of the next unit when `left` steps remain.  The first step after an upload reads raw populations (a single step).
// A unit is at most tb_steps long and at least `min_unit` (3: the in-place kernel's minimum; 4 on slabs so that the last unit
// of a call can be replayed for the one-step lag of u / rho; 2 for the two-phase kernel); what is left below that goes in
// single steps.  LBM_FLAG_EAGER_LAG (and slabs that cannot replay): the LAST step of a call is always a single step.
int unit_steps(const lbm_ctx* c, int left, bool raw) {
    if (left < 1) return 0;
    if (!c->use_tb || raw) return 1;
    const int T = c->tb_steps;
    if (is_slab(c) && !c->deep_halo && !own_transport(c)) return 1;   // (per-pass exchanges cannot be driven from outside)
    if (!lag_replayable(c, T)) {
        if (left >= T + 1) return T;
        if (T >= 3 && left - 1 >= 3) return left - 1;
        return 1;
    }
    if (T == 2) return left >= 2 ? 2 : 1;
    const int m = is_slab(c) ? 4 : 3;
    if (left >= T) {
        const int r = left - T;
        if (r == 0 || r >= m) return T;
        if (left - m >= m) return left - m;     // e.g. 8 = 4 + 4 instead of 5 + 3 singles
        return T;
    }
    return left >= m ? left : 1;
}


// ------------------------------------------------------------------------------------
// parameter checks and the plan of a context (shared by lbm_create and the dry run lbm_plan)
// ------------------------------------------------------------------------------------
// The checks of lbm_params that need no device ("" = fine).
std::string validate_params(const lbm_params* p) {
    if (!p || p->struct_size != (int32_t)sizeof(lbm_params)) return std::string("lbm_params.struct_size mismatch");
    if (p->nx < 4 || p->ny < 4) return std::string("nx, ny must be >= 4");
    if (p->y0 < 0 || p->ny_local < 2 || p->y0 + p->ny_local > p->ny) return std::string("slab rows out of range (ny_local >= 2)");
    if (p->ny_local > 65535) return std::string("ny_local > 65535 not supported");
    if (p->dtype != LBM_F32 && p->dtype != LBM_F64) return std::string("dtype must be LBM_F32 or LBM_F64");
    if (p->collision < LBM_SRT || p->collision > LBM_MRT) return std::string("collision must be SRT, TRT or MRT");
    if (p->semantics != LBM_SEM_MRT_PY && p->semantics != LBM_SEM_MRT_GPU) return std::string("bad semantics");
    if (p->turb != 0 && p->turb != 1) return std::string("turb must be 0 or 1");
    if (p->turb == 1 && p->semantics != LBM_SEM_MRT_GPU) return std::string("turb = 1 (Smagorinsky, MRT_GPU.py:368-387) exists only with MRT_GPU semantics");
    if (p->kernel < LBM_KERNEL_AUTO || p->kernel > LBM_KERNEL_STREAM) return std::string("bad kernel variant");
    if (p->kernel == LBM_KERNEL_PUSH && (p->turb || p->batch > 1 || p->y0 != 0 || p->ny_local != p->ny))
        return std::string("kernel = PUSH (the reference's two-launch scheme, for A/B) takes one whole lattice without the closure");
    if (p->layout < LBM_LAYOUT_AUTO || p->layout > LBM_LAYOUT_ROWS) return std::string("bad layout");
    if (p->batch < 0 || p->batch > 65535) return std::string("batch must be 0 .. 65535");
    if (p->arith != LBM_ARITH_STRICT && p->arith != LBM_ARITH_FAST) return std::string("arith must be LBM_ARITH_STRICT or LBM_ARITH_FAST");
    if (p->batch > 1 && (p->y0 != 0 || p->ny_local != p->ny)) return std::string("a batch of lattices cannot be slab-decomposed");
    if (p->ny_local_min < 0 || p->ny_local_min > p->ny_local) return std::string("ny_local_min must be 0 or the smallest ny_local of all ranks (<= ny_local)");
    if (p->tb_steps != 0 && (p->tb_steps < 2 || p->tb_steps > SP_MAX_S)) return std::string("tb_steps must be 0 (default) or 2 .. " + std::to_string(SP_MAX_S));
    if (p->frame_seg != 0 && p->frame_seg < 8) return std::string("frame_seg must be 0 (default) or >= 8");
    if ((p->flags & LBM_FLAG_NT_ON) && (p->flags & LBM_FLAG_NT_OFF)) return std::string("LBM_FLAG_NT_ON and LBM_FLAG_NT_OFF exclude each other");
    return std::string();
}

// Context with geometry and LAUNCH PLAN filled in from lbm_params alone -- no HIP call unless `device` (then the register counts of
// two kernels are read for the frame_beside rule).  lbm_create continues from here; lbm_plan (a dry run: what would every rank of a
// decomposition plan?) stops here.
lbm_ctx* plan_ctx(const lbm_params* p, bool device, std::string& err_out) {
    auto bail = [&](const std::string& m) -> lbm_ctx* { err_out = m; return nullptr; };
    lbm_ctx* c = new (std::nothrow) lbm_ctx();
    if (!c) return bail("out of host memory");
    c->p = *p;
    c->es = p->dtype == LBM_F32 ? 4 : 8;
    c->geo.nx = p->nx;
    c->geo.ny = p->ny_local;
    c->geo.y0 = p->y0;
    c->geo.NY = p->ny;
    c->geo.pitch = ((p->nx + 2 * GH) + 3) / 4 * 4;
    const int nplanes = p->turb ? Q + 2 : Q;   // + the two Smagorinsky history planes
    if (p->layout == LBM_LAYOUT_PLANES) {
        c->geo.plane = (long long)c->geo.pitch * (p->ny_local + 2 * GHY);
        c->geo.row = c->geo.pitch;
    } else {  // LBM_LAYOUT_ROWS (default): +10 % on the 18-stream pattern, see DESIGN.md
        c->geo.plane = c->geo.pitch;
        c->geo.row = (long long)nplanes * c->geo.pitch;
    }
    c->batch = p->batch > 1 ? p->batch : 1;
    c->bstride = (long long)nplanes * c->geo.pitch * (p->ny_local + 2 * GHY);   // lattice z of a batch starts z * bstride elements in
    const size_t bytes = (size_t)c->bstride * c->batch * c->es;
    c->lat_bytes = bytes;
    {
        // The launch plan.  Everything that shapes the exchange protocol between slabs (several steps per launch or not, how
        // many, frame width, deep halo) is derived from ny_plan = the smallest slab of the decomposition, never from this
        // rank's own share of the rows: neighbours must post matching send / receive sequences (lbm_comm_init cross-checks).
        const int ny_plan = p->ny_local_min > 0 ? p->ny_local_min : p->ny_local;
        const bool slab = p->y0 > 0 || p->y0 + p->ny_local < p->ny;
        const int V = 16 / c->es;
        const bool can_vec = p->semantics == LBM_SEM_MRT_GPU && p->nx % V == 0;
        if (p->kernel == LBM_KERNEL_VEC && !can_vec) return (delete c, bail("kernel = VEC needs MRT_GPU semantics and nx % (16 / sizeof(real)) == 0"));
        c->push = p->kernel == LBM_KERNEL_PUSH;
        c->use_vec = can_vec && p->kernel != LBM_KERNEL_GENERIC && !c->push;
        const bool can_tb = p->nx % V == 0 && p->nx >= 32 && ny_plan >= 32;
        if (p->kernel == LBM_KERNEL_TB && !can_tb) return (delete c, bail("kernel = TB needs nx % (16 / sizeof(real)) == 0, nx >= 32 and ny_local >= 32 (on every rank)"));
        // measured crossover: with one launch per frame pass (batches, LBM_FLAG_FRAME_UNFUSED) a multi-step pays from ~768^2 cells
        // (profiles/r01_logs/perf4.log); with the frame inside the tile launch a unit is ONE launch and wins from the smallest
        // lattices the in-place kernel takes (perf41.log, perf43.log: 160^2 4.1-4.5 us per step against 5.1 one step per launch)
        const bool unfused = (p->flags & LBM_FLAG_FRAME_UNFUSED) != 0;
        const bool one_launch = c->batch == 1 && !slab && !unfused;
        const bool big = one_launch ? (p->nx >= 64 && ny_plan >= 64)
                                    : (long long)p->nx * ny_plan * c->batch >= 768LL * 768LL;
        c->use_tb = can_tb && (p->kernel == LBM_KERNEL_TB || (p->kernel == LBM_KERNEL_AUTO && big));
        // Steps per launch: the in-place LDS tile kernel with S = 4 (fp32) or 3 (fp64), also with the Smagorinsky closure (its
        // history is cell-local and stays in registers).  lbm_params.tb_steps = 2..5 overrides (A/B, tests; 2 = the two-phase kernel).
        // measured in the full stepper (profiles/r01_logs/perf11.log, perf17.log, perf18.log), 4096^2 MRT: fp32 two steps 140,
        // three 176, four 207, five 207 GLUPS; fp64 two 75, three 99 (its x rim of V = 2 cells allows no more)
        // with the closure (perf23.log, 4096^2 fp32, S = 2 / 3 / 4): SRT 108 / 150 / 162, TRT 110 / 145 / 124 (S = 4 spills
        // under the 128-register occupancy floor), MRT 109 / 111 / 115; fp64 SRT 57 / 81 / 83, MRT 61 / 72 / 73
        const bool trt_turb = p->turb && p->collision == LBM_TRT;
        // arith = FAST (perf28.log, perf29.log, perf40.log): at S = 5 the strict MRT form is arithmetic-bound (209 GLUPS, as at
        // S = 4), the factored one is not: S = 3 / 4 / 5 = 184 / 227 / 277 GLUPS; SRT 189 / 237 / 250, with the closure 152 / 188 /
        // 217; MRT + closure 150 / 187 / 202; TRT 185 / 210 / 210, with the closure 152 / 163 / 156
        const bool fast = p->arith == LBM_ARITH_FAST;
        // r02: with the exact-product multiply-adds of the strict MRT operator (lbm_device.hpp) five steps pay there too
        // (profiles/r02_logs/strict_steps.log: 4096^2 fp32 226 -> 236 GLUPS, 1024^2 134 -> 145; fp64 2048^2 S = 3 / 4 / 5 = 85 / 97 / 98)
        const bool mrt_plain = p->collision == LBM_MRT && !p->turb;
        const int want32 = fast ? (p->collision == LBM_TRT ? 4 : 5) : (trt_turb ? 3 : (mrt_plain ? 5 : 4));
        // fp64 (perf46.log; an x rim of two vectors from four steps on): the factored MRT operator S = 3 / 4 / 5 = 98 / 123 / 142 GLUPS
        // at 4096^2 (8192 x 1024: 91 / 109 / 129); the strict operator is arithmetic-bound (103 / 105 / 103)
        // (SRT + closure fp64: 81 / 88 / 88 GLUPS)
        const int want64 = p->collision == LBM_MRT ? (fast || mrt_plain ? 5 : 3) : 4;
        // a lone small lattice is bound by the launch, not by arithmetic or bandwidth: more steps per launch whatever the operator
        // (perf52.log, strict: 160^2 fp32 4.33 -> 4.13 us per step with five, fp64 5.02 -> 4.65 with four)
        const bool small_lone = one_launch && (long long)p->nx * ny_plan <= 512LL * 512;
        const int want = p->tb_steps ? p->tb_steps : small_lone ? (p->dtype == LBM_F32 ? 5 : std::max(4, want64))
                                                                : (p->dtype == LBM_F32 ? want32 : want64);
        const bool deep_ok = p->nx >= 64 && ny_plan >= 64;
        // The strip-streaming kernel (lbm_stream.hpp): one workgroup per CU marches down a strip of 240 fp32 / 112 fp64 useful
        // columns, up to 8 steps per launch, no rim in y.  It needs tall segments to amortise its pipeline fill, i.e. a large
        // lattice (AUTO: below); kernel = STREAM forces it.  Between slabs the unit is an edge launch + a bulk launch (multi_step).
        const bool can_stream = can_tb && deep_ok && c->batch == 1;
        if (p->kernel == LBM_KERNEL_STREAM && !can_stream)
            return (delete c, bail("kernel = STREAM takes one lattice (no batch) with nx % (16 / sizeof(real)) == 0, nx >= 64, ny_local >= 64 (on every rank)"));
        // AUTO (profiles/r02_logs/stream_ab3.log, slab_loopback5.log; fast MRT, GLUPS stream / tile): lone 2048^2 219 / 238, 4096 x 1024 229 /
        // 240, 4096 x 2048 296 / 268, 3072^2 306 / 274, fp64 8192 x 1024 157 / 129 -> from 8 Mi cells (fp64: 2048^2 134 / 129, 2560^2 150 / 135, 4096 x
        // 1024 128 / 125 -> from 4 Mi).  A slab (its tile-kernel
        // unit is bound by the chain exchange -> frame passes -> exchange, the streaming unit is not): 4096 x 512 176 / 151,
        // 2048^2 200 / 185, 4096 x 1024 247 / 181, 4096 x 2048 280 / 233, fp64 8192 x 1024 133 / 117, 2048 x 512 110 / 96 -> from 1 Mi cells
        // and 512 rows.  Lattices
        // narrower than 2048 (few strips, not measured) keep the earlier 3072^2 rule.
        const long long cells_plan = (long long)p->nx * ny_plan;
        // r03 (profiles/r03_logs/auto_sweep.log): a lone lattice whose operator variant takes the walls inside (no frame workgroups ahead of the
        // streaming ones; waves that end with their last block) pays earlier -- stream / tile, fast MRT: fp32 4096 x 512 238 / 224, 2048^2 303 /
        // 263, 4096 x 1024 317 / 257 but 1536^2 241 / 248, 2048 x 1024 228 / 237 -> wide lattices from 2 Mi cells, the others from 4 Mi at
        // 2048 columns; fp64 1024^2 112 / 101 (strict 84.5 / 82), 2048 x 512 109 / 100, 1280^2 134 / 113, 1536^2 151 / 116 but 768^2 80 / 87 -> from
        // 1 Mi cells and 1024 columns
        const bool walls_variant = c->batch == 1 && !slab && p->semantics == LBM_SEM_MRT_GPU && !p->turb && !(p->flags & LBM_FLAG_NO_STREAM_WALLS) &&
                                   (p->collision == LBM_MRT || p->collision == LBM_SRT);
        const bool walls_pays = walls_variant && (c->es == 8 ? p->nx >= 1024 && cells_plan >= (1LL << 20)
                                                             : (p->nx >= 4096 && cells_plan >= (2LL << 20)) || (p->nx >= 2048 && cells_plan >= (4LL << 20)));
        const bool stream_pays = walls_pays ||
                                 (p->nx >= 2048 ? (slab ? cells_plan >= (1LL << 20) && ny_plan >= 512 : cells_plan >= ((c->es == 8 ? 4LL : 8LL) << 20))
                                                : cells_plan >= 3072LL * 3072);
        c->stream = can_stream && (p->kernel == LBM_KERNEL_STREAM || (p->kernel == LBM_KERNEL_AUTO && stream_pays));
        if (c->stream) {
            c->use_tb = true;
            c->tb_steps = p->tb_steps ? p->tb_steps : ST_MAX_S;
            // Frame width F, a multiple of the vector width.  Level 1 of the streaming kernel computes the cells from F - (S - 1)
            // inwards as plain pull-and-collide cells: with MRT_GPU.py's full streaming windows every cell but the wall cells
            // themselves is one (in_window), so F >= S; MRT.py's truncated windows leave kept slots in the cell next to the right /
            // bottom wall too, so F >= S + 1.
            const int fmin = p->semantics == LBM_SEM_MRT_GPU ? 0 : 1;
            c->tb_f = (c->tb_steps + fmin + 3) / 4 * 4;
            while (c->tb_steps > 2 && (p->nx < 2 * c->tb_f + 16 || ny_plan < 2 * c->tb_f + 16)) {   // (tiny lattices: keep an interior)
                c->tb_steps -= 1;
                c->tb_f = (c->tb_steps + fmin + 3) / 4 * 4;
            }
            // The wall frame of a lone lattice: inside the launch (its first workgroups; they hold a CU each for ~43 us before the
            // streaming workgroups start) or as a kernel of its own on the second stream that runs BESIDE them.  The latter needs
            // room next to a streaming workgroup, which takes all the LDS and four waves per SIMD: chosen when the registers of
            // four streaming waves and one frame wave fit the 512 per SIMD lane (allocated in blocks of 8) -- the factored
            // operators without the Smagorinsky closure -- and only in fp64, where it pays: 4096 x 4096 fast MRT 153 -> 169 GLUPS;
            // in fp32 the frame waves slow the streaming waves by more than the 43 us they save, 367 -> 338
            // (profiles/r02_logs/stream_ab18.log)
            c->tail_tiles = !slab && c->batch == 1 && c->es == 4 && p->semantics == LBM_SEM_MRT_GPU && !(p->flags & LBM_FLAG_NO_TAIL_TILES);
            // The walls inside the streaming kernel (k_stream_walls, lbm_stream.hpp): a lone lattice in MRT_GPU.py semantics needs no
            // frame -- side-wall cells in line, the lid and the bottom row as blocks of the pipeline.
            // Default: where that kernel variant needs no scratch memory (r03, 4096^2 GLUPS frame -> walls inside: fp32 MRT fast 366 -> 426, strict
            // 262 -> 292, fp64 MRT fast 171 -> 187, strict 126 -> 130, fp32 SRT 253 -> 251; the variants that spill at 128 VGPRs lose -- TRT fast 242 ->
            // 152, SRT + closure fast 256 -> 118 -- a spill reloaded behind the prefetch waits for HBM: profiles/r03_logs/walls_variants.log).  With a
            // device the kernel's scratch size is checked as well (hipFuncGetAttributes), so that a compiler that starts spilling one of
            // the chosen variants falls back to the frame instead of to half the speed (the bound, 64 B per lane, admits the strict MRT
            // operator's 52 B: a dozen registers parked once per BLOCK, outside the level loop -- what costs is a reload inside it).
            // r03, second half: a slab too (its deep halo is complete rows of the same lattice format, side-wall cells included): the edge
            // launch shrinks to the interface bands, the column strips and the lid / bottom row strip of the frame go
            const bool walls_ok = c->batch == 1 && p->semantics == LBM_SEM_MRT_GPU && (!slab || !(p->flags & LBM_FLAG_NO_DEEP_HALO));
            bool walls_pay = !p->turb && (p->collision == LBM_MRT || p->collision == LBM_SRT);
            if (walls_ok && walls_pay && device && !(p->flags & (LBM_FLAG_STREAM_WALLS | LBM_FLAG_STREAM_PAIRS))) {
                dispatch(c->p, [&](auto v) {
                    using VT = decltype(v);
                    using R = typename VT::R;
                    if constexpr (VT::SEM == SEM_GPU) {
                        hipFuncAttributes at;
                        const void* kern = slab ? reinterpret_cast<const void*>(&k_stream_walls_slab<R, VT::COLL, VT::TURB>)
                                                : reinterpret_cast<const void*>(&k_stream_walls<R, VT::COLL, VT::TURB>);
                        // (SRT: ~25 registers parked per block, outside the level loop, 96 - 120 B; 412 GLUPS fast all the same)
                        if (hipFuncGetAttributes(&at, kern) != hipSuccess || at.localSizeBytes > (p->collision == LBM_SRT ? 128u : 64u))
                            walls_pay = false;
                    }
                });
            }
            c->stream_walls = walls_ok && !(p->flags & LBM_FLAG_NO_STREAM_WALLS) &&
                              (walls_pay || (p->flags & (LBM_FLAG_STREAM_WALLS | LBM_FLAG_STREAM_PAIRS)));
            // ... and two rows per wave (k_stream_pairs): twelve waves that all work in every iteration, 10 steps per launch by default (up
            // to SP_MAX_S), a launch that costs in proportion to its steps -- so no tile-kernel tails
            c->stream_pairs = c->stream_walls && !slab && (p->flags & LBM_FLAG_STREAM_PAIRS);
            if (c->stream_pairs) {
                c->tb_steps = p->tb_steps ? p->tb_steps : 10;
                c->tail_tiles = false;
            } else if (c->tb_steps > ST_MAX_S) {
                return (delete c, bail("tb_steps " + std::to_string(ST_MAX_S + 1) + " .. " + std::to_string(SP_MAX_S) + " need the streaming kernel with two rows per wave (a lone lattice, MRT_GPU semantics)"));
            }
            // (with the walls inside the tails stay on the streaming kernel: the tile kernel's four steps are no faster any more -- 247 against 256
            // GLUPS fast, 223 / 225 strict -- and the change of kernel inside a call costs: the driver's 20 steps, repeated, 337 -> 371 GLUPS)
            if (c->stream_walls) c->tail_tiles = false;
            if (slab || c->stream_walls) c->frame_beside = false;   // (a slab's frame is its edge launch, multi_step; no frame at all with the walls inside)
            else if (p->flags & LBM_FLAG_FRAME_BESIDE_ON) c->frame_beside = true;
            else if (!(p->flags & LBM_FLAG_FRAME_BESIDE_OFF) && c->batch == 1 && c->es == 8) {
                int rs = 1 << 20, rf = 1 << 20;
                dispatch(c->p, [&](auto v) {
                    using VT = decltype(v);
                    using R = typename VT::R;
                    hipFuncAttributes at;
                    if (!device) return;
                    if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_stream<R, VT::COLL, VT::SEM, VT::TURB>)) == hipSuccess) rs = at.numRegs;
                    if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_frame_beside<R, VT::COLL, VT::SEM, VT::TURB>)) == hipSuccess) rf = at.numRegs;
                });
                c->frame_beside = (rs + 7) / 8 * 8 * 4 + (rf + 7) / 8 * 8 <= 512;
            }
        } else {
            if (want > 5) return (delete c, bail("tb_steps 6 .. " + std::to_string(SP_MAX_S) + " need kernel = STREAM (above " + std::to_string(ST_MAX_S) + ": a lone lattice in MRT_GPU semantics)"));
            c->tb_steps = want == 2 ? 2 : ((want == 4 || want == 5) && deep_ok ? want : 3);
            c->tb_f = c->tb_steps >= 4 ? 2 * TB_F : TB_F;   // F >= S + 1 and a multiple of the vector width
        }
        // tile shape of the three-step kernel, A/B in one run (profiles/r01_logs/perf14.log): 14 vectors x 28 rows beats
        // 30 x 12 by 5 % for fp32 MRT (less rim arithmetic), ties for fp64 and SRT.  (The wide variant is no longer compiled.)
        // measured (profiles/r01_logs/perf37.log, perf38.log): one launch per unit instead of S + 1 and no cross-stream dependency:
        // 4096^2 fp32 278 -> 294 GLUPS, 1024^2 fp64 67 -> 91, 1024^2 fp32 96 -> 135; a batch of 64 x 384^2 loses 5 % (its many
        // short frame workgroups do better as separate small launches), so batches keep one launch per pass
        // (r02: with the pass windows of 64-cell segments in the launch's LDS -- 76 KiB, below -- batches gain most from the fused
        // frame: 64 x 384^2 fp32 fast 132 -> 176 GLUPS aggregate, strict 120 -> 140, fp64 81 -> 86 with 24-cell segments;
        // profiles/r02_logs/batch_ab.log.  LBM_FLAG_FRAME_FUSED_BATCH is accepted and no longer needed.)
        c->frame_fused = !unfused;
        // cells of the frame per workgroup (perf43.log): short segments finish a pass in one sweep of the workgroup and suit
        // lattices whose launch is over when the frame chain is (160^2: 4.1 us per step with 16, 6.3 with 64); long ones compute
        // less overlap and suit large lattices (2048^2: 244 GLUPS with 64, 215 with 16)
        const long long cells1 = (long long)p->nx * ny_plan * c->batch;   // (what fills the device: all lattices of a batch)
        c->frame_seg = p->frame_seg ? p->frame_seg : (cells1 <= 512LL * 512 ? 16 : (cells1 <= 1024LL * 1024 ? 32 : 64));
        c->frame_lds = !(p->flags & LBM_FLAG_NO_FRAME_LDS);
        c->frame_wide = !(p->flags & LBM_FLAG_FRAME_NARROW);
        c->edge_first = !(p->flags & LBM_FLAG_NO_EDGE_FIRST);
        c->edge_reserve = !(p->flags & LBM_FLAG_NO_EDGE_RESERVE);
        c->xcd_bands = !(p->flags & LBM_FLAG_NO_XCD_BANDS);
        // a lone lattice under the tile kernel: the longest segment (in steps of 8 cells, not below 16) whose pass windows fit the
        // launch's LDS -- fp64 windows are twice the size (1024^2 fp64, five passes: 32-cell segments 85 KiB, 24-cell 69 KiB)
        if (!p->frame_seg && (one_launch || (c->batch > 1 && c->frame_fused)) && c->use_tb && !c->stream && c->tb_steps >= 3 && c->frame_lds)
            while (c->frame_seg > 16 && !frame_lds_fits(c, c->tb_steps, false, 0, TILE_FRAME_LDS_BYTES)) c->frame_seg -= 8;
        // the same for the frame workgroups inside a launch of the streaming kernel (144 KiB; fp64, eight passes: 40-cell segments):
        // 4096^2 fp64 strict 113 -> 117 GLUPS, fast with the frame inside 162 -> 171, slab 8192 x 1024 in loopback 142 -> 158
        if (!p->frame_seg && c->stream && c->frame_lds)
            while (c->frame_seg > 16 && !frame_lds_fits(c, c->tb_steps, false, 1, ST_LDS_BYTES)) c->frame_seg -= 8;
        c->deep_halo = p->semantics == LBM_SEM_MRT_GPU && !(p->flags & LBM_FLAG_NO_DEEP_HALO);
        c->use_nt = (p->flags & LBM_FLAG_NT_ON) ? true : (p->flags & LBM_FLAG_NT_OFF) ? false : (bytes > ((size_t)192 << 20));
        c->lazy_lag = !(p->flags & LBM_FLAG_EAGER_LAG);
    }
    return c;
}
}  // namespace lbmhost

using namespace lbmhost;

extern "C" {

int lbm_next_unit(const lbm_ctx* c, int steps_left) {
    if (!c || steps_left < 0) return LBM_ERR_INVALID;
    if (c->push) return steps_left > 0 ? 1 : 0;
    return unit_steps(c, steps_left, c->raw[c->cur] != 0);
}

int lbm_describe(const lbm_ctx* c, char* buf, size_t len) {
    if (!c || !buf || len == 0) return LBM_ERR_INVALID;
    const char* kern = !c->use_tb ? "none" : c->stream_pairs ? "k_stream_pairs" : c->stream_walls ? "k_stream_walls" : c->stream ? "k_stream" : c->tb_steps == 2 ? "k_step2_deep" : "k_stepS_deep";
    const int S = c->use_tb ? c->tb_steps : 1;
    long long wgs = 0, wave_updates = 0;   // per launch of S steps: workgroups of the bulk kernel; (wave, level) updates they perform
    const int V = 16 / c->es;
    if (c->stream) {
        const StreamPlan pl = plan_stream(c, S);
        wgs = (long long)pl.nstrips * pl.nsegy;
        const int nbr = (has_neighbour(c, LBM_SIDE_LOW) ? 1 : 0) + (has_neighbour(c, LBM_SIDE_HIGH) ? 1 : 0);
        const long long rows = c->geo.ny - (c->stream_walls ? nbr * c->tb_f : 2 * c->tb_f);
        // (with the walls inside a segment that starts / ends at a wall has no lead rows beyond it; a slab's bulk launch only)
        wave_updates = (long long)pl.nstrips * (rows + ((long long)pl.nsegy * 2 - (c->stream_walls ? 2 - nbr : 0)) * (S - 1)) * S;
    } else if (c->use_tb && S >= 3) {
        const int F = c->tb_f, RV = (S - 1 + V - 1) / V, TX = (16 - 2 * RV) * V, TY = 32 - 2 * (S - 1);
        const long long ntx = (c->geo.nx - 2 * F + TX - 1) / TX, nty = (c->geo.ny - 2 * F + TY - 1) / TY;
        wgs = ntx * nty * c->batch;
        long long per = 0;                 // active waves per step: rows [s - 1, 32 - (s - 1)) of 16 lanes -> (32 - 2 (s - 1)) / 4 waves
        for (int s = 1; s <= S; ++s) per += (32 - 2 * (s - 1)) / 4 + ((32 - 2 * (s - 1)) % 4 ? 1 : 0);
        wave_updates = wgs * per;
    }
    int nlat = 0;                          // device lattices held now (2 + scratch lattices in use + the lagged one): footprint = nlat * lattice_bytes
    for (int i = 0; i < NLAT; ++i) nlat += c->lat[i] ? 1 : 0;
    const int n = std::snprintf(buf, len, "kernel=%s steps_per_launch=%d frame=%d stream=%d vec=%d nt=%d deep_halo=%d frame_fused=%d lazy_lag=%d "
                                "layout=%s workgroups=%lld wave_updates=%lld cells_per_lane=%d slab=%d frame_beside=%d frame_seg=%d "
                                "lattices=%d lattice_bytes=%lld",
                                kern, S, c->use_tb ? (c->stream_walls && !is_slab(c) ? 0 : c->tb_f) : 0, c->stream ? 1 : 0, c->use_vec ? 1 : 0, c->use_nt ? 1 : 0, c->deep_halo ? 1 : 0,
                                c->frame_fused ? 1 : 0, c->lazy_lag ? 1 : 0, c->geo.row != c->geo.pitch ? "rows" : "planes", wgs, wave_updates, V,
                                is_slab(c) ? 1 : 0, c->frame_beside ? 1 : 0, c->frame_seg, nlat, (long long)c->lat_bytes);
    return n < 0 ? LBM_ERR_INVALID : (n >= (int)len ? (int)len - 1 : n);
}

// Dry run of the launch plan: what lbm_create(p) would plan and which launch units lbm_step(steps) would then run from a fresh
// lattice -- derived from lbm_params alone, NO device call (works without a GPU).  Every rank of a slab decomposition must get the
// same kernel / steps_per_launch / frame / deep_halo and the same units (they post matching send / receive sequences; lbm_comm_init
// cross-checks at run time): this lets a launcher -- and the CPU tests -- check a decomposition before any rank touches a GPU.
// ncu: compute units to plan for (0 = 256, an MI355X).  frame_beside depends on the kernels' register counts and is reported 0
// unless forced by a flag (lone lattices only: no effect on the protocol).
int lbm_plan(const lbm_params* p, int ncu, int steps, char* buf, size_t len) {
    if (!buf || len == 0) return LBM_ERR_INVALID;
    const std::string bad = validate_params(p);
    std::string perr;
    lbm_ctx* c = bad.empty() ? plan_ctx(p, false, perr) : nullptr;
    if (!c) {
        std::snprintf(buf, len, "error: %s", (bad.empty() ? perr : bad).c_str());
        return LBM_ERR_INVALID;
    }
    c->ncu = ncu > 0 ? ncu : 256;
    int n = lbm_describe(c, buf, len);
    if (n >= 0) {
        std::string u = " units=";
        int left = steps < 0 ? 0 : steps;
        bool raw = true;
        while (left > 0) {
            const int S = unit_steps(c, left, raw);
            if (S < 1) break;
            u += std::to_string(S);
            left -= S;
            if (left > 0) u += ",";
            raw = false;
        }
        if ((size_t)n + u.size() < len) { std::memcpy(buf + n, u.c_str(), u.size() + 1); n += (int)u.size(); }
    }
    delete c;
    return n;
}
}  // extern "C"
