// lbm_host.hpp -- what the host-side translation units of liblbm_hip.so share: the context (struct lbm_ctx), the lazily bound
// RCCL table, the error macros, the variant dispatcher and the prototypes of every host function that crosses a unit.
//   lbm_hip.hip     context life cycle, state upload / field export, probes              (C ABI: create .. get_tau, probes)
//   lbm_plan.hip    parameter validation, launch planning, unit sequence, the dry run    (C ABI: next_unit, describe, plan)
//   lbm_launch.hip  kernel launches and the step loop (single / multi-step units, lag)   (C ABI: step*, time_steps)
//   lbm_comm.hip    RCCL binding, halo exchanges, host-transported halos                 (C ABI: halo_*, comm_*)
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: RCCL is bound lazily with dlopen (see rccl_api)
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/lbm.h"
#include "lbm_kernels.hpp"

#include "lbm_tiles_inst.hpp"   // extern template declarations of k_stepS_deep (lbm_tiles_f32.hip / lbm_tiles_f64.hip)
#include "lbm_stream.hpp"       // ... and of k_stream, k_stream_walls, k_stream_pairs (lbm_stream*_f32.hip / _f64.hip)

// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
constexpr int LAT_LAG = 9;    // lat[LAT_LAG]: the lattice of the step before the last, recomputed on demand (lazy one-step lag)
constexpr int NLAT = 10;

struct lbm_ctx {
    lbm_params p{};
    int es = 0;  // element size
    Geo geo{};
    void* lat[NLAT] = {};       // [0], [1]: the two lattices; [2] .. [8]: frame scratch of the multi-step; [LAT_LAG]: see above
    size_t lat_bytes = 0;
    int raw[NLAT] = {1, 1, 0, 0, 0, 0, 0, 0, 0, 0};
    int cur = 0;  // lat[cur] is the source of the next step
    long long nsteps = 0;
    // One-step lag of u / rho: the fields of the last iteration are moments of the state it started from.
    // After a single step that state is still in lat[cur ^ 1] (lag = 0).  After a launch unit of S steps lat[cur ^ 1] holds the
    // state S steps back: lag = S - 1 steps are recomputed from it into lat[LAT_LAG] when lbm_get_fields / lbm_mean_u /
    // lbm_get_tau ask (lag_valid: done already) -- bit-identical, and off the path of lbm_step.
    int lag = 0;
    bool lag_valid = false;
    bool lazy_lag = true;       // (LBM_FLAG_EAGER_LAG: every lbm_step call ends with a single step instead)
    hipStream_t s_compute = nullptr, s_comm = nullptr;
    hipEvent_t ev_edges = nullptr, ev_halo = nullptr, ev_int = nullptr, ev_go = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    void* stage = nullptr;
    size_t stage_bytes = 0;
    double* red_dev = nullptr;  // lbm_mean_u: partial sums + results
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    bool thin_valid = false;    // the one-row halo of lat[cur] has been exchanged (by the RCCL path, on s_comm)
    bool frame_lds = true;      // ... keeping the intermediate passes in LDS when their windows fit (LBM_FLAG_NO_FRAME_LDS: scratch lattices)
    int frame_seg = 64;         // cells of the frame per workgroup of the fused frame passes (lbm_params.frame_seg)
    bool frame_fused = true;    // all frame passes of a multi-step in one launch (LBM_FLAG_FRAME_UNFUSED: one launch per pass)
    bool deep_halo = false;     // multi-steps between slabs exchange once per launch (MRT_GPU semantics; LBM_FLAG_NO_DEEP_HALO disables)
    bool loopback = false;      // diagnostic: 1-rank communicator, the slab exchanges halos with itself
    bool use_vec = false;       // vector kernel (MRT_GPU.py semantics, nx multiple of the vector width)
    bool use_nt = false;        // non-temporal loads/stores: lattice far larger than the 256 MiB Infinity Cache
    bool push = false;          // LBM_KERNEL_PUSH: the reference's two-launch push scheme (lat[0], lat[1]: fin ping-pong; lat[2]: ftemp)
    bool use_tb = false;        // several steps per launch (temporal blocking)
    int edge_rows = 0;          // rows next to each interface of lat[cur] that work on s_comm wrote (and s_comm's stream order therefore covers):
                                // the frame width after a multi-step unit, 1 after a single step, 0 at the start of a call (see exchange_ready)
    bool tail_tiles = false;    // streaming contexts (lone, fp32): units of 3 .. 5 steps through the tile kernel (A/B: LBM_FLAG_NO_TAIL_TILES)
    bool xcd_bands = true;      // streaming kernel: contiguous runs of segments per XCD (A/B: LBM_FLAG_NO_XCD_BANDS)
    bool edge_reserve = true;   // streaming kernel between slabs: a one-round bulk launch leaves CUs to the edge workgroups (A/B: LBM_FLAG_NO_EDGE_RESERVE)
    bool edge_first = true;     // streaming kernel between slabs: release the bulk launch behind the edge launch (A/B: LBM_FLAG_NO_EDGE_FIRST)
    bool frame_wide = true;     // frame passes through the scratch lattices: workgroups of 1024 threads (A/B: LBM_FLAG_FRAME_NARROW)
    bool frame_beside = false;  // streaming kernel of a lone lattice: the frame passes as a kernel of their own on the second stream, BESIDE the
                                // streaming workgroups (no LDS, ~70 VGPRs: fits next to them when the streaming kernel leaves registers)
    bool stream = false;        // ... by the strip-streaming kernel (lbm_stream.hpp: large lone lattices, up to 8 steps per launch)
    bool stream_walls = false;  // ... with the walls inside (k_stream_walls: a lone lattice in MRT_GPU.py semantics; no frame) (opt-in: LBM_FLAG_STREAM_WALLS)
    bool stream_pairs = false;  // ... and two rows per wave (k_stream_pairs: twelve waves, up to 10 steps per launch) (opt-in: LBM_FLAG_STREAM_PAIRS)
    int ncu = 256;              // compute units of the device (the streaming kernel runs one workgroup per CU)
    int tb_steps = 2;           // ... or three to five (in-place LDS tile kernel), up to eight (streaming kernel)
    int tb_f = TB_F;            // frame width
    int batch = 1;              // independent lattices per buffer (lbm_params.batch)
    long long bstride = 0;      // elements from one lattice of the batch to the next
    void* relax_dev = nullptr;  // batch > 1: Relax<real>[batch] on the device
    // The units of a lone lattice that need no second stream (frame and tiles / the walls inside: ONE launch on s_compute) neither wait for
    // ev_edges nor record ev_int -- two event operations per unit, 7 - 8 % of a launch-bound lattice's step (160^2: 3.51 -> 3.24 us).  Instead:
    bool int_stale = false;      // s_compute has work that ev_int does not cover yet: flush_int() before s_comm is made to wait for ev_int
    bool edges_pending = false;  // ev_edges was recorded (work on s_comm) and s_compute has not been made to wait for it since
    std::string err;
};

namespace lbmhost {

// RCCL entry points, resolved on first use.  liblbm_hip.so carries no DT_NEEDED on librccl:
// single-GPU processes never load it, and in a process that also runs torch.distributed the
// dlopen below returns the RCCL that is already mapped (same SONAME), so both share one.
struct rccl_api {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    std::string err;
};

inline int fail(lbm_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                               \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail((c), LBM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

#define NCCL_TRY(c, expr)                                                                              \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                         \
            return fail((c), LBM_ERR_COMM, std::string(#expr) + ": " + rccl().GetErrorString(r_));     \
    } while (0)

// ev_int, recorded lazily (lbm_ctx::int_stale): whoever makes s_comm wait for ev_int calls this first
inline int flush_int(lbm_ctx* c) {
    if (c->int_stale) {
        HIP_TRY(c, hipEventRecord(c->ev_int, c->s_compute));
        c->int_stale = false;
    }
    return LBM_OK;
}

template <typename R>
Relax<R> relax_of(const lbm_params& p) {
    Relax<R> w;
    w.uLB = (R)p.uLB; w.w_nu = (R)p.omega; w.w_m = (R)p.omegam;
    w.w_e = (R)p.omega_e; w.w_eps = (R)p.omega_eps; w.w_q = (R)p.omega_q;
    return w;
}

template <typename R>
Batch<R> batch_of(const lbm_ctx* c) {
    return Batch<R>{c->bstride, c->batch > 1 ? (const Relax<R>*)c->relax_dev : nullptr};
}

// output lattices of the S frame passes lat[from] -> lat[to]: the scratch lattices (null when never needed, see ensure_scratch), then lat[to]
template <typename R>
FramePtrs<R> frame_ptrs(const lbm_ctx* c, int from, int to, int S) {
    FramePtrs<R> fp;
    fp.src = (const R*)c->lat[from];
    for (int i = 0; i < 8; ++i) fp.pass[i] = i < S - 1 ? (R*)c->lat[2 + i] : (R*)c->lat[to];
    return fp;
}

inline dim3 grid_rows(const lbm_ctx* c, int nrows) { return dim3((c->geo.nx + BLK - 1) / BLK, nrows, c->batch); }

// Run-time parameters -> compile-time kernel variant (real type, collision operator, semantics, Smagorinsky).
template <typename R_, int COLL_, int SEM_, bool TURB_>
struct Variant {
    using R = R_;
    static constexpr int COLL = COLL_, SEM = SEM_;
    static constexpr bool TURB = TURB_;
};

template <typename F>
void dispatch(const lbm_params& p, F&& f) {
    auto by_sem = [&](auto real, auto coll) {
        using R = decltype(real);
        constexpr int C = decltype(coll)::value;
        constexpr int CS = C == C_MRT_FAST ? C_MRT : (C == C_SRT_FAST ? C_SRT : (C == C_TRT_FAST ? C_TRT : C));
        if (p.semantics == LBM_SEM_MRT_PY) f(Variant<R, CS, SEM_PY, false>{});   // (arith = fast: MRT_GPU semantics only)
        else if (p.turb) f(Variant<R, C, SEM_GPU, true>{});
        else f(Variant<R, C, SEM_GPU, false>{});
    };
    auto by_coll = [&](auto real) {
        switch (p.collision) {
            case LBM_SRT:
                if (p.arith == LBM_ARITH_FAST) by_sem(real, std::integral_constant<int, C_SRT_FAST>{});
                else by_sem(real, std::integral_constant<int, C_SRT>{});
                break;
            case LBM_TRT:
                if (p.arith == LBM_ARITH_FAST) by_sem(real, std::integral_constant<int, C_TRT_FAST>{});
                else by_sem(real, std::integral_constant<int, C_TRT>{});
                break;
            default:
                if (p.arith == LBM_ARITH_FAST) by_sem(real, std::integral_constant<int, C_MRT_FAST>{});
                else by_sem(real, std::integral_constant<int, C_MRT>{});
                break;
        }
    };
    if (p.dtype == LBM_F32) by_coll(float{});
    else by_coll(double{});
}
struct StreamPlan { int nstrips, nsegy, H; };

// Waiting for the device: poll for a short while, then block.  A blocking hipStreamSynchronize / hipEventSynchronize wakes the host
// tens of microseconds after the work is done -- 5 % of the driver's 20-step window of 1.1 ms (profiles/r02_logs/unit_times.log);
// a run that is still busy after SPIN_US hands the core back.
constexpr long long SPIN_US = 3000;
template <typename Q>
bool spin_until_ready(Q&& query) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = query();
        if (e == hipSuccess) return true;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); return false; }
        if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > SPIN_US) return false;
    }
}

// Is there a slab beyond this side?  Geometry decides (the slab does not touch the lid / the bottom wall there): the frame
// passes of a multi-step extend into the ghost rows of such a side whatever moves the rows -- RCCL between ranks
// (lbm_comm_init checks that rank r holds the r-th slab), the loopback diagnostic, or the caller (lbm_halo_*_rows).
inline bool has_neighbour(const lbm_ctx* c, int side) {
    return side == LBM_SIDE_LOW ? c->geo.y0 > 0 : c->geo.y0 + c->geo.ny < c->geo.NY;
}
inline bool is_slab(const lbm_ctx* c) { return has_neighbour(c, LBM_SIDE_LOW) || has_neighbour(c, LBM_SIDE_HIGH); }
// the library itself moves the halos (RCCL between ranks, or the one-GPU loopback)
inline bool own_transport(const lbm_ctx* c) { return c->comm != nullptr && (c->nranks > 1 || c->loopback); }

// Deep halo for a multi-step of S steps: the S complete rows (all planes, ghost columns included) next to each interface go to
// the neighbour's ghost rows in ONE message per side; the S frame passes then recompute a shrinking band of the neighbour's
// rows (rows -(S - i) .. for pass i) instead of exchanging one row per pass.  RCCL's latency per exchange, not its bandwidth,
// is what the per-pass scheme cannot hide: 739 KB once instead of 5 x 48 KB.  MRT_GPU semantics only: there a
// side-wall cell overwrites the slots it does not stream by the wall rule, so nothing a cell needs lives in the ghost columns
// of a ghost row (MRT.py's left wall reads parked values).
//
// The rows are described once, as blocks of contiguous elements, for RCCL (below) and for the externally driven exchange
// (lbm_halo_export_rows / lbm_halo_import_rows): [y][k][x] layout: S rows of all planes are ONE block; [k][y][x]: one per plane.
struct RowBlocks {
    int n = 0;
    char* ptr[Q + 2];
    size_t elems = 0;   // per block
};

// ---- host functions that cross a unit (defined in the unit the name of which is given) ----
// lbm_hip.hip
int ensure_stage(lbm_ctx* c, size_t bytes);
int sync_all(lbm_ctx* c);
int host_to_stage(lbm_ctx* c, const void* host, int host_dtype, int planes);
int stage_to_host(lbm_ctx* c, const void* stage, void* host, int host_dtype, int planes);
// lbm_plan.hip
bool frame_lds_fits(const lbm_ctx* c, int S, bool deep_rows, int extra = 0, long long budget = FRAME_LDS_BYTES);
int pairs_waves(int S);
StreamPlan plan_stream_on(const lbm_ctx* c, int S, int ncu, long long* cost_out);
StreamPlan plan_stream(const lbm_ctx* c, int S);
bool lag_replayable(const lbm_ctx* c, int S);
int unit_steps(const lbm_ctx* c, int left, bool raw);
std::string validate_params(const lbm_params* p);
lbm_ctx* plan_ctx(const lbm_params* p, bool device, std::string& err_out);
// lbm_launch.hip
int ensure_scratch(lbm_ctx* c, int n);
int launch_rows(lbm_ctx* c, int from, int to, int row0, int stride, int nrows, hipStream_t s);
int launch_frame(lbm_ctx* c, int from, int to, int W, hipStream_t s, int elo = 0, int ehi = 0);
int launch_frame_multi(lbm_ctx* c, int from, int to, int S, hipStream_t s, bool lo, bool hi, int extra = 0);
int launch_stream(lbm_ctx* c, int from, int to, hipStream_t s, int S, bool with_frame);
int launch_stream_edges(lbm_ctx* c, int from, int to, hipStream_t s, int S, bool lo, bool hi, int extra);
int warm_stream(lbm_ctx* c);
int launch_deep(lbm_ctx* c, int from, int to, hipStream_t s, int steps, bool with_frame = false);
void finish_unit(lbm_ctx* c, int S);
int single_step(lbm_ctx* c, bool* comm_used, bool rccl_x);
int multi_step(lbm_ctx* c, bool* comm_used, int S, bool rccl_x);
int prev_lattice(lbm_ctx* c, int* which);
int push_step(lbm_ctx* c);
int push_reset(lbm_ctx* c);
int step_many(lbm_ctx* c, int nsteps);
// lbm_comm.hip
rccl_api& rccl();
void halo_range(const lbm_ctx* c, int k, int* lo, int* hi);
const int* side_planes(int side);
char* plane_row(lbm_ctx* c, int which, int k, int y);
int enqueue_exchange(lbm_ctx* c, int which);
RowBlocks deep_blocks(lbm_ctx* c, int which, int r0, int S);
int deep_send_row0(const lbm_ctx* c, int side, int S);
int deep_recv_row0(const lbm_ctx* c, int side, int S);
int enqueue_deep_exchange(lbm_ctx* c, int which, int S);
int exchange_ready(lbm_ctx* c, int rows);
int join_comm(lbm_ctx* c);
}  // namespace lbmhost
