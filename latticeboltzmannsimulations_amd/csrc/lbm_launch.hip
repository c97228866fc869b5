// lbm_launch.hip -- kernel launches (row kernels, frame passes, tile kernel, streaming kernels) and the step loop: single-step and
// multi-step units, the lagged lattice, the overlap of halo exchange and interior work.
#include "lbm_host.hpp"

namespace lbmhost {

// Scratch lattices lat[2 .. 2 + n) of the frame passes, allocated on first use: the fused frame passes keep their
// intermediate results in LDS windows whenever those fit, so the default fp32 / fp64 paths never touch a scratch lattice and a
// streaming context holds 2 lattices (3 once the lagged fields have been asked for) instead of 2 + 7 -- 8192^2 fp64: 9.7 GB instead
// of 48.  Only frame_beside, LBM_FLAG_FRAME_UNFUSED / NO_FRAME_LDS, windows that do not fit, and the replay of the lagged lattice
// by single steps go through them.  Zeroed like the lattices themselves (dead reads of ghost positions see numbers); the wait makes
// the memory safe for either stream, once per lattice and context.
int ensure_scratch(lbm_ctx* c, int n) {
    bool fresh = false;
    for (int i = 2; i < 2 + n && i < LAT_LAG; ++i) {
        if (c->lat[i]) continue;
        hipError_t e = hipMalloc(&c->lat[i], c->lat_bytes);
        if (e != hipSuccess) {
            c->lat[i] = nullptr;
            return fail(c, LBM_ERR_NOMEM, std::string("hipMalloc(scratch lattice of the frame passes): ") + hipGetErrorString(e));
        }
        HIP_TRY(c, hipMemsetAsync(c->lat[i], 0, c->lat_bytes, c->s_compute));
        fresh = true;
    }
    if (fresh) HIP_TRY(c, hipStreamSynchronize(c->s_compute));
    return LBM_OK;
}

// One single step, lat[from] -> lat[to], on local rows row0 + i*stride, i in [0, nrows).
int launch_rows(lbm_ctx* c, int from, int to, int row0, int stride, int nrows, hipStream_t s) {
    if (nrows <= 0) return LBM_OK;
    dispatch(c->p, [&](auto v) {
        using VT = decltype(v);
        using R = typename VT::R;
        const R* src = (const R*)c->lat[from];
        R* dst = (R*)c->lat[to];
        const int raw = c->raw[from];
        if (VT::SEM == SEM_GPU && c->use_vec) {
            constexpr int V = 16 / (int)sizeof(R);
            const int nxb = (c->geo.nx / V + BLK - 1) / BLK, nblocks = nxb * nrows;
            if (c->use_nt)
                hipLaunchKernelGGL((k_step_vec<R, VT::COLL, V, true, VT::TURB>), dim3(nblocks, c->batch), dim3(BLK), 0, s, src, dst, c->geo,
                                   relax_of<R>(c->p), batch_of<R>(c), raw, row0, stride, nxb, nblocks);
            else
                hipLaunchKernelGGL((k_step_vec<R, VT::COLL, V, false, VT::TURB>), dim3(nblocks, c->batch), dim3(BLK), 0, s, src, dst, c->geo,
                                   relax_of<R>(c->p), batch_of<R>(c), raw, row0, stride, nxb, nblocks);
        } else {
            hipLaunchKernelGGL((k_step_generic<R, VT::COLL, VT::SEM, VT::TURB>), grid_rows(c, nrows), dim3(BLK), 0, s, src, dst,
                               c->geo, relax_of<R>(c->p), batch_of<R>(c), raw, row0, stride);
        }
    });
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

// One single step on the frame of width W, lat[from] -> lat[to] (one pass of a multi-step; never a raw lattice).
int launch_frame(lbm_ctx* c, int from, int to, int W, hipStream_t s, int elo, int ehi) {
    dispatch(c->p, [&](auto v) {
        using VT = decltype(v);
        using R = typename VT::R;
        constexpr int V = 16 / (int)sizeof(R);
        const int vec_rows = VT::SEM == SEM_GPU && c->use_vec && c->geo.nx % V == 0 ? 1 : 0;   // row strips by vector cells
        const long long cells = (2LL * W + elo + ehi) * (vec_rows ? c->geo.nx / V : c->geo.nx) + 2LL * W * (c->geo.ny - 2 * W);
        hipLaunchKernelGGL((k_step_frame<R, VT::COLL, VT::SEM, VT::TURB>), dim3((unsigned)((cells + BLK - 1) / BLK), c->batch), dim3(BLK), 0, s,
                           (const R*)c->lat[from], (R*)c->lat[to], c->geo, relax_of<R>(c->p), batch_of<R>(c), W, elo, ehi, vec_rows);
    });
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

// extra: rows of the neighbours' side that the row strips own on top of the slab's (see frame_passes)
int launch_frame_multi(lbm_ctx* c, int from, int to, int S, hipStream_t s, bool lo, bool hi, int extra) {
    const bool beside = c->frame_beside && !lo && !hi && c->batch == 1;
    if (beside || !frame_lds_fits(c, S, lo || hi, extra)) {
        const int rc = ensure_scratch(c, S - 1);
        if (rc) return rc;
    }
    dispatch(c->p, [&](auto v) {
        using VT = decltype(v);
        using R = typename VT::R;
        const FramePtrs<R> fp = frame_ptrs<R>(c, from, to, S);
        const int F = c->tb_f, L = c->frame_seg, nsegx = (c->geo.nx + L - 1) / L, nsegy = (c->geo.ny - 2 * F + L - 1) / L;
        if (beside) {
            hipLaunchKernelGGL((k_frame_beside<R, VT::COLL, VT::SEM, VT::TURB>), dim3(2 * nsegx + 2 * nsegy), dim3(BLK), 0, s, fp, c->geo, relax_of<R>(c->p), F, S,
                               nsegx, nsegy, L);
            return;
        }
        const bool in_lds = frame_lds_fits(c, S, lo || hi, extra);
        if (!in_lds && c->frame_wide)
            hipLaunchKernelGGL((k_frame_multi<R, VT::COLL, VT::SEM, VT::TURB, 1024>), dim3(2 * nsegx + 2 * nsegy, c->batch), dim3(1024), 0, s, fp, c->geo,
                               relax_of<R>(c->p), batch_of<R>(c), F, S, nsegx, nsegy, lo ? 1 + extra : 0, hi ? 1 + extra : 0, L, 0);
        else
            hipLaunchKernelGGL((k_frame_multi<R, VT::COLL, VT::SEM, VT::TURB, BLK>), dim3(2 * nsegx + 2 * nsegy, c->batch), dim3(BLK), 0, s, fp, c->geo,
                               relax_of<R>(c->p), batch_of<R>(c), F, S, nsegx, nsegy, lo ? 1 + extra : 0, hi ? 1 + extra : 0, L, in_lds ? 1 : 0);
    });
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

int launch_stream(lbm_ctx* c, int from, int to, hipStream_t s, int S, bool with_frame) {
    if (c->stream_walls) {
        // walls included, no frame at all: the whole lattice in one launch (with_frame), or a slab's rows between its edge bands
        const bool slab = is_slab(c);
        if (slab == with_frame) return fail(c, LBM_ERR_STATE, "internal: the streaming kernel with the walls inside takes a whole lone lattice or a slab's bulk rows");
        const int ybeg = has_neighbour(c, LBM_SIDE_LOW) ? c->tb_f : 0, yend = c->geo.ny - (has_neighbour(c, LBM_SIDE_HIGH) ? c->tb_f : 0);
        dispatch(c->p, [&](auto v) {
            using VT = decltype(v);
            using R = typename VT::R;
            if constexpr (VT::SEM == SEM_GPU) {
                const StreamPlan pl = plan_stream(c, S);
                if (c->stream_pairs)
                    hipLaunchKernelGGL((k_stream_pairs<R, VT::COLL, VT::TURB>), dim3(pl.nstrips * pl.nsegy), dim3(64 * pairs_waves(S)), 0, s,
                                       (const R*)c->lat[from], (R*)c->lat[to], c->geo, relax_of<R>(c->p), S, pl.nstrips, pl.H, c->xcd_bands ? 1 : 0);
                else if (slab)
                    hipLaunchKernelGGL((k_stream_walls_slab<R, VT::COLL, VT::TURB>), dim3(pl.nstrips * pl.nsegy), dim3(ST_NT), 0, s, (const R*)c->lat[from],
                                       (R*)c->lat[to], c->geo, relax_of<R>(c->p), S, pl.nstrips, pl.H, c->xcd_bands ? 1 : 0, ybeg, yend, 0, 0, 0, 0);
                else
                    hipLaunchKernelGGL((k_stream_walls<R, VT::COLL, VT::TURB>), dim3(pl.nstrips * pl.nsegy), dim3(ST_NT), 0, s, (const R*)c->lat[from],
                                       (R*)c->lat[to], c->geo, relax_of<R>(c->p), S, pl.nstrips, pl.H, c->xcd_bands ? 1 : 0);
            }
        });
        HIP_TRY(c, hipGetLastError());
        return LBM_OK;
    }
    const bool use_lds = frame_lds_fits(c, S, false, 0, ST_LDS_BYTES);
    if (with_frame && !use_lds) {
        const int rc = ensure_scratch(c, S - 1);
        if (rc) return rc;
    }
    dispatch(c->p, [&](auto v) {
        using VT = decltype(v);
        using R = typename VT::R;
        const int F = c->tb_f, xe = c->geo.nx - F, ye = c->geo.ny - F;
        const StreamPlan pl = plan_stream(c, S);
        const FramePtrs<R> fp = frame_ptrs<R>(c, from, to, S);
        const int L = c->frame_seg, nsegx = (c->geo.nx + L - 1) / L, nsegy = (c->geo.ny - 2 * F + L - 1) / L;
        const int nframe = with_frame ? 2 * nsegx + 2 * nsegy : 0;
        hipLaunchKernelGGL((k_stream<R, VT::COLL, VT::SEM, VT::TURB>), dim3(nframe + pl.nstrips * pl.nsegy), dim3(ST_NT), 0, s,
                           (const R*)c->lat[from], (R*)c->lat[to], c->geo, relax_of<R>(c->p), S, F, xe, ye, pl.nstrips, pl.H,
                           fp, nframe, nsegx, nsegy, L, use_lds ? 1 : 0, 0, 0, 0, c->xcd_bands ? 1 : 0);
    });
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

// The edge launch of a slab's unit under the streaming kernel: everything but the bulk rows [F, ny - F) x [F, nx - F) -- the wall
// frame (the column strips over the slab's whole height, the row strip of a lid / bottom wall this slab holds) by the frame
// workgroups and, on each side with a neighbour, the F rows next to the interface between the column strips as a short
// streaming segment that starts in the neighbour's rows of the deep halo.  It writes every row the next exchange sends.
// extra: rows of the neighbours' side owned on top (1 for the lagged lattice, see frame_passes).
int launch_stream_edges(lbm_ctx* c, int from, int to, hipStream_t s, int S, bool lo, bool hi, int extra) {
    if (c->stream_walls) {   // the walls inside: the edge launch is the interface bands alone, over the whole width (side-wall cells in line)
        dispatch(c->p, [&](auto v) {
            using VT = decltype(v);
            using R = typename VT::R;
            if constexpr (VT::SEM == SEM_GPU) {
                const StreamPlan pl = plan_stream(c, S);
                hipLaunchKernelGGL((k_stream_walls_slab<R, VT::COLL, VT::TURB>), dim3(pl.nstrips * ((lo ? 1 : 0) + (hi ? 1 : 0))), dim3(ST_NT), 0, s,
                                   (const R*)c->lat[from], (R*)c->lat[to], c->geo, relax_of<R>(c->p), S, pl.nstrips, pl.H, 0, 0, 0,
                                   (lo ? 1 : 0) | (hi ? 2 : 0), lo ? 1 + extra : 0, hi ? 1 + extra : 0, c->tb_f);
            }
        });
        HIP_TRY(c, hipGetLastError());
        return LBM_OK;
    }
    const bool use_lds = frame_lds_fits(c, S, false, extra, ST_LDS_BYTES);
    if (!use_lds) {
        const int rc = ensure_scratch(c, S - 1);
        if (rc) return rc;
    }
    dispatch(c->p, [&](auto v) {
        using VT = decltype(v);
        using R = typename VT::R;
        const int F = c->tb_f, xe = c->geo.nx - F, ye = c->geo.ny - F;
        const StreamPlan pl = plan_stream(c, S);
        const FramePtrs<R> fp = frame_ptrs<R>(c, from, to, S);
        const int bands = (lo ? 1 : 0) | (hi ? 2 : 0);
        const int ybeg = lo ? -extra : F, yend = hi ? c->geo.ny + extra : c->geo.ny - F;
        const int L = c->frame_seg, nsegx = (c->geo.nx + L - 1) / L, nsegy = (yend - ybeg + L - 1) / L;
        const int nframe = 2 * nsegx + 2 * nsegy;
        hipLaunchKernelGGL((k_stream<R, VT::COLL, VT::SEM, VT::TURB>), dim3(nframe + pl.nstrips * ((lo ? 1 : 0) + (hi ? 1 : 0))), dim3(ST_NT), 0, s,
                           (const R*)c->lat[from], (R*)c->lat[to], c->geo, relax_of<R>(c->p), S, F, xe, ye, pl.nstrips, pl.H,
                           fp, nframe, nsegx, nsegy, L, use_lds ? 1 : 0, lo ? 1 + extra : 0,
                           hi ? 1 + extra : 0, bands, 0);
    });
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

// The first launch of the streaming kernel in a process costs ~1.4 ms (code upload, 144 KiB of LDS, scratch set-up).  Where the first
// units of a run may go to the tile kernel (tail_tiles) that cost would land in the middle of a run -- in the driver's 20 timed
// steps after a 5-step warm-up, for one -- so lbm_create pays it: one workgroup that returns at once (its segment is empty).
int warm_stream(lbm_ctx* c) {
    dispatch(c->p, [&](auto v) {
        using VT = decltype(v);
        using R = typename VT::R;
        if constexpr (VT::SEM == SEM_GPU) {
            if (c->stream_walls) {   // (H = 0: the one workgroup's segment is empty)
                if (c->stream_pairs)
                    hipLaunchKernelGGL((k_stream_pairs<R, VT::COLL, VT::TURB>), dim3(1), dim3(64 * pairs_waves(c->tb_steps)), 0, c->s_compute, (const R*)c->lat[0],
                                       (R*)c->lat[1], c->geo, relax_of<R>(c->p), c->tb_steps, 1, 0, 0);
                else if (is_slab(c))   // (rows [0, 0): empty)
                    hipLaunchKernelGGL((k_stream_walls_slab<R, VT::COLL, VT::TURB>), dim3(1), dim3(ST_NT), 0, c->s_compute, (const R*)c->lat[0], (R*)c->lat[1],
                                       c->geo, relax_of<R>(c->p), c->tb_steps, 1, 0, 0, 0, 0, 0, 0, 0, 0);
                else
                    hipLaunchKernelGGL((k_stream_walls<R, VT::COLL, VT::TURB>), dim3(1), dim3(ST_NT), 0, c->s_compute, (const R*)c->lat[0], (R*)c->lat[1],
                                       c->geo, relax_of<R>(c->p), c->tb_steps, 1, 0, 0);
                return;
            }
        }
        const int F = c->tb_f;
        const FramePtrs<R> fp = frame_ptrs<R>(c, 0, 1, 1);
        hipLaunchKernelGGL((k_stream<R, VT::COLL, VT::SEM, VT::TURB>), dim3(1), dim3(ST_NT), 0, c->s_compute, (const R*)c->lat[0], (R*)c->lat[1],
                           c->geo, relax_of<R>(c->p), c->tb_steps, F, c->geo.nx - F, /*ye=*/F, 1, 1, fp, 0, 1, 1, c->frame_seg, 0, 0, 0, 0, 0);
    });
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

int launch_deep(lbm_ctx* c, int from, int to, hipStream_t s, int steps, bool with_frame) {
    // A short unit of a lone fp32 lattice (the tail of a call: 3 .. 5 steps) goes to the tile kernel: a launch of the streaming
    // kernel costs nearly the same whatever its length (4096^2 fast: 311 us for four steps, 374 for eight), the tile kernel's four
    // steps take ~290 (strict ~300 against ~350): the driver's 20 timed steps, fast 1088 -> 1066 us, strict 1466 -> 1412
    // (profiles/r02_logs/tail_tiles.log)
    const bool tail = c->stream && c->tail_tiles && c->frame_fused && with_frame && steps >= 3 && steps <= 5;
    if (c->stream && !tail) return launch_stream(c, from, to, s, steps, with_frame);
    const int S_tile = steps >= 3 ? (steps == 4 || steps == 5 ? steps : 3) : 2;
    const bool tile_frame_lds = frame_lds_fits(c, S_tile, false, 0, TILE_FRAME_LDS_BYTES);
    if (with_frame && steps >= 3 && !tile_frame_lds) {
        const int rc = ensure_scratch(c, S_tile - 1);
        if (rc) return rc;
    }
    dispatch(c->p, [&](auto v) {
        using VT = decltype(v);
        using R = typename VT::R;
        const int F = c->tb_f, xe = c->geo.nx - F, ye = c->geo.ny - F;
        if (steps >= 3) {
            constexpr int V = 16 / (int)sizeof(R);
            auto go = [&](auto steps, auto wide) {
                constexpr int S = decltype(steps)::value;
                constexpr bool WIDE = decltype(wide)::value;
                constexpr int PVC = WIDE ? 32 : 16, RV = (S - 1 + V - 1) / V, TX = (PVC - 2 * RV) * V, TY = 512 / PVC - 2 * (S - 1);
                const int ntx = (xe - F + TX - 1) / TX, nty = (ye - F + TY - 1) / TY;
                const FramePtrs<R> fp = frame_ptrs<R>(c, from, to, S);
                const int L = c->frame_seg, nsegx = (c->geo.nx + L - 1) / L, nsegy = (c->geo.ny - 2 * F + L - 1) / L;
                const int nframe = with_frame ? 2 * nsegx + 2 * nsegy : 0;
                hipLaunchKernelGGL((k_stepS_deep<R, VT::COLL, VT::SEM, S, WIDE, VT::TURB>), dim3(nframe + ntx * nty, c->batch), dim3(512), 0, s,
                                   (const R*)c->lat[from], (R*)c->lat[to], c->geo, relax_of<R>(c->p), batch_of<R>(c), F, xe, ye, ntx, ntx * nty,
                                   fp, nframe, nsegx, nsegy, L, tile_frame_lds ? 1 : 0);
            };
            {   // (fp64: the x rim of S >= 4 is two vectors wide)
                if (steps == 4) { go(std::integral_constant<int, 4>{}, std::false_type{}); return; }
                if (steps == 5) { go(std::integral_constant<int, 5>{}, std::false_type{}); return; }
            }
            go(std::integral_constant<int, 3>{}, std::false_type{});
            return;
        }
        constexpr int V = 16 / (int)sizeof(R), TX = tb_txv<VT::TURB>() * V, TY = tb_ty<VT::TURB>();
        const int ntx = (xe - TB_F + TX - 1) / TX, nty = (ye - TB_F + TY - 1) / TY;   // two steps: F = TB_F
        hipLaunchKernelGGL((k_step2_deep<R, VT::COLL, VT::TURB>), dim3(ntx * nty, c->batch), dim3(TB_NT), 0, s, (const R*)c->lat[from],
                           (R*)c->lat[to], c->geo, relax_of<R>(c->p), batch_of<R>(c), xe, ye, ntx, ntx * nty);
    });
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

// bookkeeping after a launch unit of S steps lat[cur] -> lat[cur ^ 1]
void finish_unit(lbm_ctx* c, int S) {
    c->cur ^= 1;
    c->raw[c->cur] = c->push ? 1 : 0;   // (push scheme: the lattices hold plain populations, nothing to stream at read time)
    c->nsteps += S;
    c->lag = S - 1;
    c->lag_valid = false;
    c->thin_valid = false;
}

// Every launch unit (one single step or one multi-step) of a slab follows one protocol on the two streams:
//   s_comm    (highest priority): [the unit's halo exchange -- RCCL, or nothing when the caller has moved the rows] ->
//                                 waits ev_int (bulk kernel of the previous unit) -> wall / slab-edge work of this unit ->
//                                 records ev_edges;
//   s_compute                   : waits ev_edges of the PREVIOUS unit, runs the bulk kernel, records ev_int.
// The exchange of a unit is enqueued first: it only touches rows that the edge / frame kernels of the previous unit wrote
// (same stream, in order) and ghost rows, so it runs beside the previous unit's bulk kernel (exchange_ready() adds the wait for
// s_compute where that premise does not hold); the small kernels run beside the bulk kernel of the same unit.  Nothing is carried from one unit to the next except thin_valid (a one-row halo that is
// already in place, e.g. the one lbm_step leaves for lbm_get_fields).
int single_step(lbm_ctx* c, bool* comm_used, bool rccl_x) {
    const int ny = c->geo.ny, a = c->cur, b = c->cur ^ 1;
    if (is_slab(c)) {
        // edge rows 0 and ny-1 (they read the ghost rows) | interior rows
        if (rccl_x && !c->raw[a] && !c->thin_valid) {   // (a raw lattice is not streamed: no halo needed)
            int rc = exchange_ready(c, 1);
            if (rc == LBM_OK) rc = enqueue_exchange(c, a);
            if (rc) return rc;
        }
        int rc = flush_int(c);
        if (rc) return rc;
        HIP_TRY(c, hipStreamWaitEvent(c->s_comm, c->ev_int, 0));
        rc = launch_rows(c, a, b, 0, ny - 1, 2, c->s_comm);
        if (rc) return rc;
        HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_edges, 0));
        HIP_TRY(c, hipEventRecord(c->ev_edges, c->s_comm));
        c->edges_pending = true;
        rc = launch_rows(c, a, b, 1, 1, ny - 2, c->s_compute);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_int, c->s_compute));
        finish_unit(c, 1);
        c->edge_rows = 1;
        *comm_used = true;
        return LBM_OK;
    }
    if (c->edges_pending) {   // frame kernels of an earlier multi-step
        HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_edges, 0));
        c->edges_pending = false;
    }
    int rc = launch_rows(c, a, b, 0, 1, ny, c->s_compute);
    if (rc) return rc;
    c->int_stale = true;
    finish_unit(c, 1);
    return LBM_OK;
}

// S steps: lat[a] (state n) -> lat[b] (state n+S).  Bulk: the deep-interior kernel on cells >= tb_f away
// from walls and slab edges.  Frame: S ordinary single steps on strips of decreasing width (tb_f + S - i for pass i; pass i+1
// pulls from one cell further out than it writes), through the scratch lattices, the last one into lat[b].
// Between slabs the frame passes and the exchanges share the second stream, beside the tile kernel; under the streaming kernel the
// frame work of a slab is its edge launch (launch_stream_edges: column strips + the interface rows as short streaming segments).  With the deep halo
// (MRT_GPU semantics) the row strips of pass i start S - i rows inside the neighbour's rows received before the unit, and that
// is the unit's only exchange; otherwise every pass but the last is followed by a one-row exchange.  (Running row and
// column strips as separate launches on separate streams was measured and lost 8 %: profiles/r01_logs/perf31.log, perf35.log.)
int multi_step(lbm_ctx* c, bool* comm_used, int S, bool rccl_x) {
    const bool slab = is_slab(c);
    if (!slab && ((c->stream_walls && S >= 2) || (S >= 3 && c->frame_fused && !(c->stream && c->frame_beside)))) {   // a lone lattice: frame and tiles in ONE launch, everything on the compute stream
        if (c->edges_pending) {   // (frame launches of an earlier unit on the second stream, if any)
            HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_edges, 0));
            c->edges_pending = false;
        }
        int rc = launch_deep(c, c->cur, c->cur ^ 1, c->s_compute, S, true);
        if (rc) return rc;
        c->int_stale = true;      // (ev_int is recorded when something on s_comm is made to wait for it: flush_int)
        finish_unit(c, S);
        return LBM_OK;
    }
    const bool deep = slab && c->deep_halo;
    if (slab && !deep && !rccl_x) return fail(c, LBM_ERR_STATE, "a multi-step unit of a slab needs the deep halo (MRT_GPU semantics) or the in-library exchange");
    const int a = c->cur, b = c->cur ^ 1;
    int rc;
    if (slab && rccl_x && (deep || !c->thin_valid)) {
        rc = exchange_ready(c, deep ? S : 1);
        if (rc == LBM_OK) rc = deep ? enqueue_deep_exchange(c, a, S) : enqueue_exchange(c, a);
        if (rc) return rc;
    }
    rc = flush_int(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamWaitEvent(c->s_comm, c->ev_int, 0));
    HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_edges, 0));   // this unit's tile kernel needs the previous unit's frame
    int from = a;
    const bool has_lo = has_neighbour(c, LBM_SIDE_LOW), has_hi = has_neighbour(c, LBM_SIDE_HIGH);
    if ((c->frame_fused || c->stream_walls) && S >= 3 && c->stream && deep) {
        // The streaming kernel between slabs: the edge launch (interface rows + column strips, everything the next exchange sends)
        // here, the bulk launch below.  Both become ready when the previous bulk launch ends, and the bulk launch -- one
        // workgroup per CU for its whole run -- must not take the CUs first: the edge workgroups would run last, and the next
        // exchange after them, in the open.  So the bulk launch is released from THIS stream, one cross-stream hop behind the
        // edge launch -- when the bulk launch runs more than one round of workgroups (16384 x 2048 fp32 slab in loopback 319 -> 359
        // GLUPS, 8192 x 1024 fp64 133 -> 142); a one-round launch does not gain and a short one loses (4096 x 4096 355 -> 351,
        // 4096 x 1024 249 -> 205: profiles/r02_logs/slab_loopback7.log).
        //
        // Why the release (ev_go) and the early exchange cannot break an ordering -- unit n goes lat[a] -> lat[b]; E = exchange, G = edge
        // launch, B = bulk launch; s_comm runs  E_n, wait(ev_int: B_{n-1}), [record ev_go], G_n, record ev_edges;  s_compute runs
        // wait(ev_edges: G_{n-1}), [wait ev_go], B_n, record ev_int:
        //   * E_n sends rows [0, S) / [ny - S, ny) of lat[a] and fills lat[a]'s ghost rows.  The rows it sends lie inside the F >= S edge
        //     rows G_{n-1} wrote -- same stream, earlier -- unless the previous unit was no streaming unit: then edge_rows < S and
        //     exchange_ready() makes s_comm wait for ev_int first.  Nothing else touches those rows or lat[a]'s ghost rows meanwhile:
        //     B_{n-1}, which may still run, writes lat[a]'s rows [F, ny - F) only and reads lat[b].
        //   * G_n reads lat[a] up to F + S - 1 rows from an interface plus the ghost rows: written by G_{n-1} and E_n (same stream,
        //     earlier) and by B_{n-1} (the wait on ev_int sits between E_n and G_n).  It writes lat[b]'s edge rows, last read by
        //     G_{n-1} / E_{n-1} (same stream, earlier) and by B_{n-1} (waited for).
        //   * B_n reads lat[a]'s rows from F - (S - 1) on: B_{n-1}'s (same stream) and G_{n-1}'s (the wait on ev_edges, recorded after
        //     G_{n-1}).  It writes lat[b]'s rows [F, ny - F): last read by B_{n-1} (same stream) and G_{n-1} (waited for).  The next
        //     exchange E_{n+1}, which may run beside B_n, touches lat[b]'s edge and ghost rows only -- disjoint from B_n's.
        //   * ev_go only ADDS an edge: B_n after everything s_comm had enqueued when it was recorded (E_n and the wait for B_{n-1}).  It is
        //     recorded (host order) before s_compute is told to wait for it, and what it waits for -- ev_int of unit n - 1 -- was recorded
        //     on s_compute before that wait: no cycle, no wait on an event not yet recorded.  Its price: B_n also waits for E_n, which it
        //     does not need; E_n has had the whole of B_{n-1} to finish, so this costs only when a neighbour is that late -- and then G_n,
        //     which B_{n+1} needs, waits for the same exchange anyway.
        // (the events are re-recorded every unit: a wait refers to the last record before it in host order -- the one named above)
        const StreamPlan pl = plan_stream(c, S);
        if (c->edge_first && (long long)pl.nstrips * pl.nsegy > c->ncu) {
            HIP_TRY(c, hipEventRecord(c->ev_go, c->s_comm));
            HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_go, 0));
        }
        rc = launch_stream_edges(c, a, b, c->s_comm, S, has_lo, has_hi, 0);
        if (rc) return rc;
    } else if (c->frame_fused && S >= 3 && (!slab || deep)) {
        rc = launch_frame_multi(c, a, b, S, c->s_comm, deep && has_lo, deep && has_hi);
        if (rc) return rc;
    } else {
      rc = ensure_scratch(c, 2);
      if (rc) return rc;
      for (int i = 1; i <= S; ++i) {
        const int to = i == S ? b : 2 + ((i - 1) & 1);
        const int ext = deep ? S - i : 0;
        rc = launch_frame(c, from, to, c->tb_f + S - i, c->s_comm, has_lo ? ext : 0, has_hi ? ext : 0);
        if (rc) return rc;
        if (slab && !deep && i < S) {
            rc = enqueue_exchange(c, to);
            if (rc) return rc;
        }
        from = to;
      }
    }
    HIP_TRY(c, hipEventRecord(c->ev_edges, c->s_comm));
    c->edges_pending = true;
    rc = launch_deep(c, a, b, c->s_compute, S);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_int, c->s_compute));
    c->int_stale = false;
    finish_unit(c, S);
    c->edge_rows = c->tb_f;
    *comm_used = true;
    return LBM_OK;
}

// Recompute the lattice of the step before the last into lat[LAT_LAG] (see lbm_ctx::lag); returns the index of the lattice
// whose gathered populations are the state the LAST iteration started from.
int prev_lattice(lbm_ctx* c, int* which) {
    if (c->nsteps == 0) { *which = c->cur; return LBM_OK; }
    if (c->lag == 0) { *which = c->cur ^ 1; return LBM_OK; }
    *which = LAT_LAG;
    if (c->lag_valid) return LBM_OK;
    if (!c->lat[LAT_LAG]) {
        hipError_t e = hipMalloc(&c->lat[LAT_LAG], c->lat_bytes);
        if (e != hipSuccess) return fail(c, LBM_ERR_NOMEM, std::string("hipMalloc(lag lattice): ") + hipGetErrorString(e));
        HIP_TRY(c, hipMemsetAsync(c->lat[LAT_LAG], 0, c->lat_bytes, c->s_compute));
    }
    const int k = c->lag, from = c->cur ^ 1;
    const bool slab = is_slab(c);
    int rc;
    if (k >= 3 && c->tb_steps >= 3) {   // one multi-step launch of k steps (a slab: from the deep halo still in lat[from]'s ghost rows)
        if (!slab && (c->frame_fused || c->stream_walls)) {
            rc = launch_deep(c, from, LAT_LAG, c->s_compute, k, true);
        } else {
            const bool lo = has_neighbour(c, LBM_SIDE_LOW), hi = has_neighbour(c, LBM_SIDE_HIGH);
            // one row more than a launch unit computes: the field export pulls the slab's first / last row from the first ghost
            // rows of this lattice (the unit received S = k + 1 rows per side: enough)
            if ((c->frame_fused || c->stream_walls) && c->stream && c->deep_halo) rc = launch_stream_edges(c, from, LAT_LAG, c->s_compute, k, lo, hi, 1);
            else if (c->frame_fused) rc = launch_frame_multi(c, from, LAT_LAG, k, c->s_compute, lo, hi, 1);
            else {
                rc = ensure_scratch(c, 2);
                int f = from;
                for (int i = 1; i <= k && rc == LBM_OK; ++i) {
                    const int to = i == k ? LAT_LAG : 2 + ((i - 1) & 1);
                    rc = launch_frame(c, f, to, c->tb_f + k - i, c->s_compute, lo ? k - i + 1 : 0, hi ? k - i + 1 : 0);
                    f = to;
                }
            }
            if (rc == LBM_OK) rc = launch_deep(c, from, LAT_LAG, c->s_compute, k);
        }
        if (rc) return rc;
    } else {                            // k single steps (a lone lattice), through scratch lattice 2
        if (slab) return fail(c, LBM_ERR_STATE, "internal: the last unit of a slab cannot be replayed");
        if (k > 1 && (rc = ensure_scratch(c, 2)) != LBM_OK) return rc;
        int f = from;
        for (int i = 1; i <= k; ++i) {
            const int to = i == k ? LAT_LAG : (f == 2 ? 3 : 2);
            c->raw[to] = 0;
            rc = launch_rows(c, f, to, 0, 1, c->geo.ny, c->s_compute);
            if (rc) return rc;
            f = to;
        }
    }
    c->raw[LAT_LAG] = 0;
    c->lag_valid = true;
    return LBM_OK;
}

// One step of the push scheme: collide-and-push lat[cur] -> ftemp (lat[2]); wall rules on ftemp + copy -> lat[cur ^ 1].
int push_step(lbm_ctx* c) {
    dispatch(c->p, [&](auto v) {
        using VT = decltype(v);
        using R = typename VT::R;
        const dim3 g = grid_rows(c, c->geo.ny);
        hipLaunchKernelGGL((k_push_collide<R, VT::COLL, VT::SEM>), g, dim3(BLK), 0, c->s_compute, (const R*)c->lat[c->cur], (R*)c->lat[2], c->geo,
                           relax_of<R>(c->p));
        hipLaunchKernelGGL((k_push_bc<R, VT::COLL, VT::SEM>), g, dim3(BLK), 0, c->s_compute, (const R*)c->lat[c->cur], (R*)c->lat[2],
                           (R*)c->lat[c->cur ^ 1], c->geo, (R)c->p.uLB);
    });
    HIP_TRY(c, hipGetLastError());
    finish_unit(c, 1);
    return LBM_OK;
}

// ftemp starts as a copy of fin (MRT_GPU.py:324)
int push_reset(lbm_ctx* c) {
    if (!c->push) return LBM_OK;
    HIP_TRY(c, hipMemcpyAsync(c->lat[2], c->lat[0], (size_t)c->bstride * c->es, hipMemcpyDeviceToDevice, c->s_compute));
    return LBM_OK;
}

int step_many(lbm_ctx* c, int nsteps) {
    if (c->push) {
        for (int i = 0; i < nsteps; ++i) {
            int rc = push_step(c);
            if (rc) return rc;
        }
        return LBM_OK;
    }
    const bool slab = is_slab(c);
    if (slab && !own_transport(c))
        return fail(c, LBM_ERR_STATE, "lbm_step on a slab without a communicator: its ghost rows would never be exchanged (attach one with "
                                      "lbm_comm_init, or drive the slab with lbm_step_edges/interior/finish, lbm_step_unit and the lbm_halo_* calls)");
    bool comm_used = false;
    if (c->use_tb || slab)   // (a lone lattice stepping one step per launch uses one stream, no events)
        HIP_TRY(c, hipEventRecord(c->ev_int, c->s_compute));   // everything enqueued so far (init, upload, earlier calls)
    c->edge_rows = 0;        // the first exchange of the call waits for it
    int left = nsteps;
    while (left > 0) {
        const int S = unit_steps(c, left, c->raw[c->cur] != 0);
        const int rc = S > 1 ? multi_step(c, &comm_used, S, true) : single_step(c, &comm_used, true);
        if (rc) return rc;
        left -= S;
    }
    if (slab && nsteps > 0 && !c->raw[c->cur]) {   // the populations lbm_get_fields returns for the slab's first / last row need the one-row halo
        int rc = exchange_ready(c, 1);
        if (rc == LBM_OK) rc = enqueue_exchange(c, c->cur);
        if (rc) return rc;
        c->thin_valid = true;
        comm_used = true;
    }
    if (comm_used) return join_comm(c);
    return LBM_OK;
}
}  // namespace lbmhost

using namespace lbmhost;

extern "C" {

int lbm_step(lbm_ctx* c, int nsteps) {
    if (!c || nsteps < 0) return fail(c, LBM_ERR_INVALID, "lbm_step: bad argument");
    HIP_TRY(c, hipSetDevice(c->p.device));
    return step_many(c, nsteps);
}

int lbm_time_steps(lbm_ctx* c, int nsteps, double* ms) {
    if (!c || nsteps < 0 || !ms) return fail(c, LBM_ERR_INVALID, "lbm_time_steps: bad argument");
    HIP_TRY(c, hipSetDevice(c->p.device));
    HIP_TRY(c, hipEventRecord(c->ev_t0, c->s_compute));
    int rc = step_many(c, nsteps);   // (joins the communication stream into s_compute before returning)
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_t1, c->s_compute));
    if (!spin_until_ready([&] { return hipEventQuery(c->ev_t1); })) HIP_TRY(c, hipEventSynchronize(c->ev_t1));
    float f = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&f, c->ev_t0, c->ev_t1));
    *ms = (double)f;
    return LBM_OK;
}

int lbm_step_edges(lbm_ctx* c) {
    if (!c) return LBM_ERR_INVALID;
    if (c->push) return fail(c, LBM_ERR_STATE, "the split-step calls do not apply to kernel = PUSH");
    HIP_TRY(c, hipSetDevice(c->p.device));
    return launch_rows(c, c->cur, c->cur ^ 1, 0, c->geo.ny - 1, 2, c->s_compute);
}

int lbm_step_interior(lbm_ctx* c) {
    if (!c) return LBM_ERR_INVALID;
    if (c->push) return fail(c, LBM_ERR_STATE, "the split-step calls do not apply to kernel = PUSH");
    HIP_TRY(c, hipSetDevice(c->p.device));
    return launch_rows(c, c->cur, c->cur ^ 1, 1, 1, c->geo.ny - 2, c->s_compute);
}

int lbm_step_finish(lbm_ctx* c) {
    if (!c) return LBM_ERR_INVALID;
    if (c->push) return fail(c, LBM_ERR_STATE, "the split-step calls do not apply to kernel = PUSH");
    finish_unit(c, 1);
    return LBM_OK;
}

int lbm_step_unit(lbm_ctx* c, int S) {
    if (!c) return LBM_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->p.device));
    if (c->push || !c->use_tb) return fail(c, LBM_ERR_STATE, "lbm_step_unit: this context steps one step per launch (lbm_next_unit() is 1)");
    if (own_transport(c)) return fail(c, LBM_ERR_STATE, "lbm_step_unit: a communicator is attached, lbm_step() moves the halos itself");
    if (c->raw[c->cur]) return fail(c, LBM_ERR_STATE, "lbm_step_unit: the first step after an upload is a single step");
    const bool ok = c->tb_steps == 2 ? S == 2 : (S >= 3 && S <= c->tb_steps);
    if (!ok)
        return fail(c, LBM_ERR_INVALID, c->tb_steps == 2 ? std::string("lbm_step_unit: this context runs units of 2 steps")
                                                         : "lbm_step_unit: unit_steps must be 3 .. " + std::to_string(c->tb_steps) +
                                                               " (this context's steps per launch; lbm_next_unit plans 4 or more on a slab)");
    if (is_slab(c) && !c->deep_halo) return fail(c, LBM_ERR_STATE, "lbm_step_unit on a slab needs the deep halo (MRT_GPU semantics)");
    bool comm_used = false;
    HIP_TRY(c, hipEventRecord(c->ev_int, c->s_compute));   // everything enqueued so far, the imported rows included
    c->edge_rows = 0;
    int rc = multi_step(c, &comm_used, S, false);
    if (rc) return rc;
    if (comm_used) return join_comm(c);
    return LBM_OK;
}
}  // extern "C"
