// lbm_hip.hip -- liblbm_hip.so: kernels, context and the C ABI declared in include/lbm.h.
// gfx950 only.  See DESIGN.md for the data layout and the per-kernel roofline notes.
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

// ------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------
#include "lbm_tiles_inst.hpp"   // extern template declarations of k_stepS_deep unless LBM_SINGLE_TU
#include "lbm_stream.hpp"       // ... and of k_stream (lbm_stream_f32.hip / lbm_stream_f64.hip)

// (not a template: defined in this translation unit only)
__global__ __launch_bounds__(BLK) void k_copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * BLK + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * BLK;
    for (; i < n; i += stride) b[i] = a[i];
}

// fp32 fused multiply-add rate probe (lbm_fma_rate): 16 independent packed FMAs per thread and iteration, nothing else
__global__ __launch_bounds__(BLK) void k_fma_burn(float* __restrict__ sink, int iters, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + (float)i, seed - (float)threadIdx.x * 1e-6f};
    const f2 m = f2{0.999f, 1.001f}, b = f2{1e-3f, -1e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, b);
    }
    f2 t = a[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += a[i];
    if (t.x + t.y == 123456.789f) sink[blockIdx.x * BLK + threadIdx.x] = t.x;   // (never true in practice: keeps the loop alive)
}

// out[z] = scale * (partial[z * nper + 0] + partial[z * nper + 1] + ...), added in index order by one thread per lattice
__global__ __launch_bounds__(BLK) void k_reduce_final(const double* __restrict__ partial, int nper, int nlat, double scale, double* __restrict__ out) {
    const int z = blockIdx.x * BLK + threadIdx.x;
    if (z >= nlat) return;
    double a = 0.0;
    for (int i = 0; i < nper; ++i) a += partial[(size_t)z * nper + i];
    out[z] = scale * a;
}

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
    std::string err;
};

namespace {

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

rccl_api& rccl() {
    static rccl_api a;
    if (a.ok || !a.err.empty()) return a;
    void* h = nullptr;
    // r02's exit abort ("double free or corruption (!prev)", rc 134, when a process created a communicator here and imported torch
    // afterwards) -- cause, from two backtraces (profiles/r03_logs/rc134_gdb.log, rc134_gdb2.log): the abort is in exit(), in the
    // destructor of the namespace-scope  std::map<amd::smi::DevInfoTypes, const char*>  that librocm_smi64 AND libamd_smi both define
    // (the same sources, one default-visibility symbol).  This function used to dlopen RCCL with RTLD_GLOBAL, which puts RCCL's
    // dependency librocm_smi64 into the process's GLOBAL symbol scope; a library with the same symbol that is mapped later --
    // the PyTorch wheel's librocm_smi64 (soname .7, beside /opt/rocm's .1), or /opt/rocm's libamd_smi.so, which `import torch` pulls
    // in -- then binds its own static initialiser and its own atexit destructor to the FIRST definition: one object, constructed twice,
    // destroyed twice.  With torch imported first nothing was global and each library kept its own copy.  Fix, in the library (a C
    // caller is covered too): RCCL is opened RTLD_LOCAL -- its entry points are taken with dlsym from the handle anyway -- so nothing
    // of its dependency chain can be interposed on.  On top of that, ONE ROCm stack per process: (1) a copy of RCCL that is already
    // mapped, under either name (PyTorch bundles "librccl.so", ROCm installs "librccl.so.1"); (2) else the RCCL that sits next to the
    // HIP runtime THIS process runs on (the loader says where hipGetDeviceCount lives: a Python process with PyTorch installed runs on
    // the wheel's bundled libamdhip64, _lib.py preloads it; a C caller on /opt/rocm's) -- its $ORIGIN rpath keeps its whole
    // dependency chain in that installation; (3) else by name.
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : {"librccl.so", "librccl.so.1"}) {
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (h) break;
    }
    if (!h) {
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char* n : {"librccl.so", "librccl.so.1"}) {
                    h = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL);
                    if (h) break;
                }
            }
        }
    }
    for (const char* n : names) {
        if (h) break;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) { a.err = std::string("dlopen(librccl): ") + dlerror(); return a; }
#define RCCL_SYM(field, sym)                                                        \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, #sym));                   \
    if (!a.field) { a.err = std::string("dlsym(" #sym ") failed"); return a; }
    RCCL_SYM(GetUniqueId, ncclGetUniqueId)
    RCCL_SYM(CommInitRank, ncclCommInitRank)
    RCCL_SYM(CommDestroy, ncclCommDestroy)
    RCCL_SYM(GroupStart, ncclGroupStart)
    RCCL_SYM(GroupEnd, ncclGroupEnd)
    RCCL_SYM(Send, ncclSend)
    RCCL_SYM(Recv, ncclRecv)
    RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef RCCL_SYM
    a.ok = true;
    return a;
}

int fail(lbm_ctx* c, int code, const std::string& msg) {
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

// output lattices of the S frame passes lat[from] -> lat[to]: the scratch lattices (null when never needed, see ensure_scratch), then lat[to]
template <typename R>
FramePtrs<R> frame_ptrs(const lbm_ctx* c, int from, int to, int S) {
    FramePtrs<R> fp;
    fp.src = (const R*)c->lat[from];
    for (int i = 0; i < 8; ++i) fp.pass[i] = i < S - 1 ? (R*)c->lat[2 + i] : (R*)c->lat[to];
    return fp;
}

dim3 grid_rows(const lbm_ctx* c, int nrows) { return dim3((c->geo.nx + BLK - 1) / BLK, nrows, c->batch); }

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
int launch_frame(lbm_ctx* c, int from, int to, int W, hipStream_t s, int elo = 0, int ehi = 0) {
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

// All S frame passes lat[from] -> lat[to] in one launch (k_frame_multi); lo / hi: the slab has a neighbour below row 0 / above
// row ny - 1 whose rows lie in the ghost rows (deep halo).
// Do the LDS windows of the fused frame passes fit (two buffers of the largest pass-1 rectangle plus its ring)?
bool frame_lds_fits(const lbm_ctx* c, int S, bool deep_rows, int extra = 0, long long budget = FRAME_LDS_BYTES) {
    if (!c->frame_lds) return false;
    const int F = c->tb_f, L = c->frame_seg, m = S - 1, np = c->p.turb ? Q + 2 : Q;
    const long long row_strip = (long long)(L + 2 * m + 2) * (F + m + (deep_rows ? m + extra : 0) + 2);
    const long long col_strip = (long long)(F + m + 2) * (L + 2 * m + extra + 2);
    return 2 * np * std::max(row_strip, col_strip) * c->es <= budget;
}

// extra: rows of the neighbours' side that the row strips own on top of the slab's (see frame_passes)
int launch_frame_multi(lbm_ctx* c, int from, int to, int S, hipStream_t s, bool lo, bool hi, int extra = 0) {
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

// `steps` steps on the deep interior, lat[from] -> lat[to] (c->tb_steps, or fewer for the last launch of a call).
// Segments of the streaming kernel: strips of 64 V - 2 R useful columns, each cut into nsegy segments of H rows; one
// workgroup per segment and ONE workgroup per CU, so the plan minimises rounds x iterations per segment (a segment of H rows
// takes H + 2 (S - 1) rows through the pipeline plus its fill).
bool has_neighbour(const lbm_ctx* c, int side);
bool is_slab(const lbm_ctx* c);
struct StreamPlan { int nstrips, nsegy, H; };
// waves per workgroup of k_stream_pairs for S steps per launch: two idle pair-slots to load the next pair in
int pairs_waves(int S) { return std::min(SP_MAX_WAVES, S + 2); }
StreamPlan plan_stream_on(const lbm_ctx* c, int S, int ncu, long long* cost_out) {
    const int V = 16 / c->es, Rr = stream_rim(S, V), TXu = 64 * V - 2 * Rr, F = c->stream_walls ? 0 : c->tb_f;
    const int cols = c->geo.nx - 2 * F, rows = c->geo.ny - 2 * F;
    // (with the walls inside: strips over the whole width, no rim at a wall; segments over the whole height)
    StreamPlan best{c->stream_walls ? stream_walls_strips(c->geo.nx, S, V) : (cols + TXu - 1) / TXu, 1, rows};
    long long best_cost = -1;
    for (int n = 1; n <= 256 && n * 8 <= std::max(rows, 8); ++n) {
        const int H = (rows + n - 1) / n, nseg = (rows + H - 1) / H;
        const long long segs = (long long)best.nstrips * nseg, rounds = (segs + ncu - 1) / ncu;
        long long iters = (H + 2 * (S - 1) + ST_WAVES - 1) / ST_WAVES * ST_WAVES + ST_WAVES;
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
    if (!(is_slab(c) && c->deep_halo && c->frame_fused && c->edge_reserve) || S < 3) return p0;
    if ((long long)p0.nstrips * p0.nsegy > c->ncu) return p0;     // several rounds: the bulk launch is released behind the edge launch instead
    const int nb = (has_neighbour(c, LBM_SIDE_LOW) ? 1 : 0) + (has_neighbour(c, LBM_SIDE_HIGH) ? 1 : 0), L = c->frame_seg;
    const long long n_edge = (long long)nb * p0.nstrips + 2LL * ((c->geo.ny + L - 1) / L) + (2LL - nb) * ((c->geo.nx + L - 1) / L);
    const long long edge_it = (c->tb_f + 2 * (S - 1) + ST_WAVES - 1) / ST_WAVES * ST_WAVES + ST_WAVES;
    if (2 * cost0 > 3 * edge_it) return p0;   // (measured: a bulk launch longer than ~1.5 edge workgroups overlaps them well enough as it is)
    StreamPlan best = p0;
    long long best_cost = cost0 + edge_it * ((n_edge + c->ncu - 1) / c->ncu);
    for (int div = 1; div <= 3; ++div) {
        const long long r = (n_edge + div - 1) / div;
        if (r < 1 || r > c->ncu / 2) continue;
        long long cb = 0;
        const StreamPlan p = plan_stream_on(c, S, c->ncu - (int)r, &cb);
        const long long cost = std::max(cb, edge_it * ((n_edge + r - 1) / r));
        if (cost < best_cost) { best_cost = cost; best = p; }
    }
    return best;
}

int launch_stream(lbm_ctx* c, int from, int to, hipStream_t s, int S, bool with_frame) {
    if (c->stream_walls) {   // the whole lattice, walls included, in one launch of the streaming kernel: no frame at all
        if (!with_frame) return fail(c, LBM_ERR_STATE, "internal: the streaming kernel with the walls inside takes the whole lattice");
        dispatch(c->p, [&](auto v) {
            using VT = decltype(v);
            using R = typename VT::R;
            if constexpr (VT::SEM == SEM_GPU) {
                const StreamPlan pl = plan_stream(c, S);
                if (c->stream_pairs)
                    hipLaunchKernelGGL((k_stream_pairs<R, VT::COLL, VT::TURB>), dim3(pl.nstrips * pl.nsegy), dim3(64 * pairs_waves(S)), 0, s,
                                       (const R*)c->lat[from], (R*)c->lat[to], c->geo, relax_of<R>(c->p), S, pl.nstrips, pl.H, c->xcd_bands ? 1 : 0);
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

int launch_deep(lbm_ctx* c, int from, int to, hipStream_t s, int steps, bool with_frame = false) {
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

// x-range [lo, hi] of plane k that a slab neighbour actually pulls from a halo row
void halo_range(const lbm_ctx* c, int k, int* lo, int* hi) {
    const int X = c->p.nx, cx = cxk(k);
    int d0, d1;  // destination window in x
    if (c->p.semantics == LBM_SEM_MRT_PY) {
        d0 = cx > 0 ? 1 : 0;
        d1 = cx > 0 ? X - 2 : (cx < 0 ? X - 3 : X - 1);
    } else {
        d0 = cx > 0 ? 1 : 0;
        d1 = cx < 0 ? X - 2 : X - 1;
    }
    *lo = d0 - cx;
    *hi = d1 - cx;
}

// planes leaving through a side: LOW (towards smaller y): cy = +1 -> k = 2, 5, 6;
// HIGH (towards larger y): cy = -1 -> k = 4, 7, 8
const int* side_planes(int side) {
    static const int low[3] = {2, 5, 6}, high[3] = {4, 7, 8};
    return side == LBM_SIDE_LOW ? low : high;
}

char* plane_row(lbm_ctx* c, int which, int k, int y) {
    return (char*)c->lat[which] + ((size_t)k * c->geo.plane + (size_t)c->geo.at(0, y)) * c->es;
}

int ensure_stage(lbm_ctx* c, size_t bytes) {
    if (c->stage_bytes >= bytes) return LBM_OK;
    if (c->stage) { (void)hipFree(c->stage); c->stage = nullptr; c->stage_bytes = 0; }
    hipError_t e = hipMalloc(&c->stage, bytes);
    if (e != hipSuccess) return fail(c, LBM_ERR_NOMEM, std::string("hipMalloc(staging): ") + hipGetErrorString(e));
    c->stage_bytes = bytes;
    return LBM_OK;
}

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
int sync_all(lbm_ctx* c) {
    if (!spin_until_ready([&] { return hipStreamQuery(c->s_compute); })) HIP_TRY(c, hipStreamSynchronize(c->s_compute));
    if (!spin_until_ready([&] { return hipStreamQuery(c->s_comm); })) HIP_TRY(c, hipStreamSynchronize(c->s_comm));
    return LBM_OK;
}

// Is there a slab beyond this side?  Geometry decides (the slab does not touch the lid / the bottom wall there): the frame
// passes of a multi-step extend into the ghost rows of such a side whatever moves the rows -- RCCL between ranks
// (lbm_comm_init checks that rank r holds the r-th slab), the loopback diagnostic, or the caller (lbm_halo_*_rows).
bool has_neighbour(const lbm_ctx* c, int side) {
    return side == LBM_SIDE_LOW ? c->geo.y0 > 0 : c->geo.y0 + c->geo.ny < c->geo.NY;
}
bool is_slab(const lbm_ctx* c) { return has_neighbour(c, LBM_SIDE_LOW) || has_neighbour(c, LBM_SIDE_HIGH); }
// the library itself moves the halos (RCCL between ranks, or the one-GPU loopback)
bool own_transport(const lbm_ctx* c) { return c->comm != nullptr && (c->nranks > 1 || c->loopback); }

#ifdef LBM_DEBUG
bool debug_skip_exchange() {   // timing diagnostic of debug builds only: results between slabs are wrong
    static const bool skip = std::getenv("LBM_DEBUG_SKIP_EXCHANGE") != nullptr;
    return skip;
}
#else
constexpr bool debug_skip_exchange() { return false; }
#endif

// RCCL exchange of the one-row halo of lat[which] with both neighbours, on s_comm
int enqueue_exchange(lbm_ctx* c, int which) {
    if (debug_skip_exchange()) return LBM_OK;
    const ncclDataType_t dt = c->p.dtype == LBM_F32 ? ncclFloat : ncclDouble;
    const int ny = c->geo.ny;
    NCCL_TRY(c, rccl().GroupStart());
    for (int side = 0; side < 2; ++side) {
        int peer = side == LBM_SIDE_LOW ? c->rank - 1 : c->rank + 1;
        if (c->loopback) peer = 0;                 // the slab is its own neighbour
        if (!has_neighbour(c, side)) continue;
        // planes leaving / arriving through this side.  In loopback mode both sides talk to rank 0, and RCCL
        // pairs the i-th send to a peer with the i-th receive from it: what leaves through the OTHER side is
        // sent here, so that the HIGH row lands in the LOW ghost row and vice versa (periodic wrap).
        const int sside = c->loopback ? (side ^ 1) : side;
        const int* out = side_planes(sside);
        const int* in = side_planes(side ^ 1);
        const int send_row = sside == LBM_SIDE_LOW ? 0 : ny - 1;
        const int recv_row = side == LBM_SIDE_LOW ? -1 : ny;
        for (int j = 0; j < 3; ++j) {
            int lo, hi;
            halo_range(c, out[j], &lo, &hi);
            NCCL_TRY(c, rccl().Send(plane_row(c, which, out[j], send_row) + (size_t)lo * c->es, (size_t)(hi - lo + 1), dt,
                                 peer, c->comm, c->s_comm));
            halo_range(c, in[j], &lo, &hi);
            NCCL_TRY(c, rccl().Recv(plane_row(c, which, in[j], recv_row) + (size_t)lo * c->es, (size_t)(hi - lo + 1), dt,
                                 peer, c->comm, c->s_comm));
        }
    }
    NCCL_TRY(c, rccl().GroupEnd());
    return LBM_OK;
}

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
RowBlocks deep_blocks(lbm_ctx* c, int which, int r0, int S) {
    RowBlocks b;
    const int nplanes = c->p.turb ? Q + 2 : Q;
    const bool rows_layout = c->geo.row != c->geo.pitch;
    auto at = [&](int k) { return (char*)c->lat[which] + ((size_t)k * c->geo.plane + (size_t)(r0 + GHY) * c->geo.row) * c->es; };
    if (rows_layout) {
        b.n = 1; b.ptr[0] = at(0); b.elems = (size_t)S * c->geo.row;
    } else {
        b.n = nplanes; b.elems = (size_t)S * c->geo.pitch;
        for (int k = 0; k < nplanes; ++k) b.ptr[k] = at(k);
    }
    return b;
}
int deep_send_row0(const lbm_ctx* c, int side, int S) { return side == LBM_SIDE_LOW ? 0 : c->geo.ny - S; }
int deep_recv_row0(const lbm_ctx* c, int side, int S) { return side == LBM_SIDE_LOW ? -S : c->geo.ny; }

int enqueue_deep_exchange(lbm_ctx* c, int which, int S) {
    if (debug_skip_exchange()) return LBM_OK;
    const ncclDataType_t dt = c->p.dtype == LBM_F32 ? ncclFloat : ncclDouble;
    NCCL_TRY(c, rccl().GroupStart());
    for (int side = 0; side < 2; ++side) {
        int peer = side == LBM_SIDE_LOW ? c->rank - 1 : c->rank + 1;
        if (c->loopback) peer = 0;
        if (!has_neighbour(c, side)) continue;
        const int sside = c->loopback ? (side ^ 1) : side;   // see enqueue_exchange
        const RowBlocks snd = deep_blocks(c, which, deep_send_row0(c, sside, S), S);
        const RowBlocks rcv = deep_blocks(c, which, deep_recv_row0(c, side, S), S);
        for (int i = 0; i < snd.n; ++i) {
            NCCL_TRY(c, rccl().Send(snd.ptr[i], snd.elems, dt, peer, c->comm, c->s_comm));
            NCCL_TRY(c, rccl().Recv(rcv.ptr[i], rcv.elems, dt, peer, c->comm, c->s_comm));
        }
    }
    NCCL_TRY(c, rccl().GroupEnd());
    return LBM_OK;
}

// An exchange is enqueued on s_comm AHEAD of the wait for the previous bulk kernel, so that it runs beside it.  That is only right if
// the rows it sends were written by work on s_comm itself: the previous unit's frame passes / edge launch (F rows) or edge kernel
// (one row).  The S rows of a deep exchange after a SINGLE step, and any exchange at the start of a call (the lattice may come from
// an upload, an import or a recomputation on s_compute), must wait for s_compute first -- found by a soak of several solvers in
// one process (tools/soak.py seq: the first solver of a process was slow enough to hide it; profiles/r02_logs/soak_bisect2.log).
int exchange_ready(lbm_ctx* c, int rows) {
#ifdef LBM_DEBUG
    static const bool off = std::getenv("LBM_DEBUG_NO_EXCHANGE_READY") != nullptr;   // (debug builds: shows that the tests see the race)
    if (off) return LBM_OK;
#endif
    if (c->edge_rows < rows) HIP_TRY(c, hipStreamWaitEvent(c->s_comm, c->ev_int, 0));
    return LBM_OK;
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
        HIP_TRY(c, hipStreamWaitEvent(c->s_comm, c->ev_int, 0));
        int rc = launch_rows(c, a, b, 0, ny - 1, 2, c->s_comm);
        if (rc) return rc;
        HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_edges, 0));
        HIP_TRY(c, hipEventRecord(c->ev_edges, c->s_comm));
        rc = launch_rows(c, a, b, 1, 1, ny - 2, c->s_compute);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_int, c->s_compute));
        finish_unit(c, 1);
        c->edge_rows = 1;
        *comm_used = true;
        return LBM_OK;
    }
    if (c->use_tb) HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_edges, 0));   // frame kernels of an earlier multi-step
    int rc = launch_rows(c, a, b, 0, 1, ny, c->s_compute);
    if (rc) return rc;
    if (c->use_tb) HIP_TRY(c, hipEventRecord(c->ev_int, c->s_compute));
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
        HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_edges, 0));   // (frame launches of an earlier unit, if any)
        int rc = launch_deep(c, c->cur, c->cur ^ 1, c->s_compute, S, true);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_int, c->s_compute));
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
    HIP_TRY(c, hipStreamWaitEvent(c->s_comm, c->ev_int, 0));
    HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_edges, 0));   // this unit's tile kernel needs the previous unit's frame
    int from = a;
    const bool has_lo = has_neighbour(c, LBM_SIDE_LOW), has_hi = has_neighbour(c, LBM_SIDE_HIGH);
    if (c->frame_fused && S >= 3 && c->stream && deep) {
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
    rc = launch_deep(c, a, b, c->s_compute, S);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_int, c->s_compute));
    finish_unit(c, S);
    c->edge_rows = c->tb_f;
    *comm_used = true;
    return LBM_OK;
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
            if (c->frame_fused && c->stream && c->deep_halo) rc = launch_stream_edges(c, from, LAT_LAG, c->s_compute, k, lo, hi, 1);
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

// later single-stream work (export, timing event, externally driven calls) must see the s_comm results
int join_comm(lbm_ctx* c) {
    HIP_TRY(c, hipEventRecord(c->ev_halo, c->s_comm));
    HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_halo, 0));
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

// host <-> staging helpers.  Host arrays are whole-lattice [planes][nx][NY]; staging is
// [planes][nx][ny_local].
template <typename D, typename S>
void convert_rows(D* dst, const S* src, size_t planes_nx, int ny, int NY, int y0, bool to_host) {
    for (size_t r = 0; r < planes_nx; ++r)
        for (int y = 0; y < ny; ++y) {
            if (to_host) dst[r * NY + y0 + y] = (D)src[r * ny + y];
            else dst[r * ny + y] = (D)src[r * NY + y0 + y];
        }
}

int host_to_stage(lbm_ctx* c, const void* host, int host_dtype, int planes) {
    const int nx = c->geo.nx, ny = c->geo.ny, NY = c->geo.NY, y0 = c->geo.y0;
    const size_t rows = (size_t)planes * nx;
    if (host_dtype == c->p.dtype) {
        HIP_TRY(c, hipMemcpy2D(c->stage, (size_t)ny * c->es, (const char*)host + (size_t)y0 * c->es, (size_t)NY * c->es,
                               (size_t)ny * c->es, rows, hipMemcpyHostToDevice));
        return LBM_OK;
    }
    std::vector<char> tmp(rows * ny * c->es);
    if (c->p.dtype == LBM_F32) convert_rows((float*)tmp.data(), (const double*)host, rows, ny, NY, y0, false);
    else convert_rows((double*)tmp.data(), (const float*)host, rows, ny, NY, y0, false);
    HIP_TRY(c, hipMemcpy(c->stage, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    return LBM_OK;
}

int stage_to_host(lbm_ctx* c, const void* stage, void* host, int host_dtype, int planes) {
    const int nx = c->geo.nx, ny = c->geo.ny, NY = c->geo.NY, y0 = c->geo.y0;
    const size_t rows = (size_t)planes * nx;
    if (host_dtype == c->p.dtype) {
        HIP_TRY(c, hipMemcpy2D((char*)host + (size_t)y0 * c->es, (size_t)NY * c->es, stage, (size_t)ny * c->es,
                               (size_t)ny * c->es, rows, hipMemcpyDeviceToHost));
        return LBM_OK;
    }
    std::vector<char> tmp(rows * ny * c->es);
    HIP_TRY(c, hipMemcpy(tmp.data(), stage, tmp.size(), hipMemcpyDeviceToHost));
    if (c->p.dtype == LBM_F32) convert_rows((double*)host, (const float*)tmp.data(), rows, ny, NY, y0, true);
    else convert_rows((float*)host, (const double*)tmp.data(), rows, ny, NY, y0, true);
    return LBM_OK;
}

// grid of the host-layout <-> lattice kernels: tiles of trx<R>() columns x 32 rows
template <typename R>
dim3 grid_tiles(const lbm_ctx* c) { return dim3((c->geo.nx + trx<R>() - 1) / trx<R>(), (c->geo.ny + 31) / 32, c->batch); }

template <typename R>
int export_fin_t(lbm_ctx* c) {
    const dim3 g = grid_tiles<R>(c);
    const R* src = (const R*)c->lat[c->cur];
    if (c->p.semantics == LBM_SEM_MRT_PY)
        hipLaunchKernelGGL((k_export_fin<R, SEM_PY>), g, dim3(BLK), 0, c->s_compute, src, c->geo, c->raw[c->cur], (R)c->p.uLB, (R*)c->stage, c->bstride);
    else
        hipLaunchKernelGGL((k_export_fin<R, SEM_GPU>), g, dim3(BLK), 0, c->s_compute, src, c->geo, c->raw[c->cur], (R)c->p.uLB, (R*)c->stage, c->bstride);
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

template <typename R>
int export_macro_t(lbm_ctx* c) {
    const dim3 g = grid_tiles<R>(c);
    // the fields of the LAST iteration are the moments of the state that iteration started
    // from, i.e. of the previous lattice (prev_lattice: still intact after a single step, recomputed after a multi-step unit)
    int which = 0;
    int rc = prev_lattice(c, &which);
    if (rc) return rc;
    const R* src = (const R*)c->lat[which];
    if (c->p.semantics == LBM_SEM_MRT_PY)
        hipLaunchKernelGGL((k_export_macro<R, SEM_PY>), g, dim3(BLK), 0, c->s_compute, src, c->geo, c->raw[which], (R)c->p.uLB, (R*)c->stage, c->bstride);
    else
        hipLaunchKernelGGL((k_export_macro<R, SEM_GPU>), g, dim3(BLK), 0, c->s_compute, src, c->geo, c->raw[which], (R)c->p.uLB, (R*)c->stage, c->bstride);
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

template <typename R>
int export_tau_t(lbm_ctx* c) {
    int which = 0;
    int rc = prev_lattice(c, &which);
    if (rc) return rc;
    const R* src = (const R*)c->lat[which];
    if (c->p.arith == LBM_ARITH_FAST)
        hipLaunchKernelGGL((k_export_tau<R, true>), grid_tiles<R>(c), dim3(BLK), 0, c->s_compute, src, c->geo, c->raw[which], relax_of<R>(c->p), batch_of<R>(c), c->p.turb, (R*)c->stage);
    else
        hipLaunchKernelGGL((k_export_tau<R, false>), grid_tiles<R>(c), dim3(BLK), 0, c->s_compute, src, c->geo, c->raw[which], relax_of<R>(c->p), batch_of<R>(c), c->p.turb, (R*)c->stage);
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

constexpr int RED_BLOCKS = 1024;   // partial sums per lattice of lbm_mean_u

template <typename R>
int reduce_u_t(lbm_ctx* c) {
    int which = 0;
    int rc = prev_lattice(c, &which);
    if (rc) return rc;
    const R* src = (const R*)c->lat[which];
    const dim3 g(RED_BLOCKS, 1, c->batch);
    if (c->p.semantics == LBM_SEM_MRT_PY)
        hipLaunchKernelGGL((k_reduce_u<R, SEM_PY>), g, dim3(BLK), 0, c->s_compute, src, c->geo, c->raw[which], (R)c->p.uLB, c->bstride, c->red_dev);
    else
        hipLaunchKernelGGL((k_reduce_u<R, SEM_GPU>), g, dim3(BLK), 0, c->s_compute, src, c->geo, c->raw[which], (R)c->p.uLB, c->bstride, c->red_dev);
    HIP_TRY(c, hipGetLastError());
    return LBM_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
// The checks of lbm_params that need no device ("" = fine).
static std::string validate_params(const lbm_params* p) {
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
static lbm_ctx* plan_ctx(const lbm_params* p, bool device, std::string& err_out) {
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
        const bool stream_pays = p->nx >= 2048 ? (slab ? cells_plan >= (1LL << 20) && ny_plan >= 512 : cells_plan >= ((c->es == 8 ? 4LL : 8LL) << 20))
                                               : cells_plan >= 3072LL * 3072;
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
            // the chosen variants falls back to the frame instead of to half the speed.
            const bool walls_ok = !slab && c->batch == 1 && p->semantics == LBM_SEM_MRT_GPU;
            bool walls_pay = !p->turb && (p->collision == LBM_MRT || (p->collision == LBM_SRT && c->es == 8));
            if (walls_ok && walls_pay && device && !(p->flags & (LBM_FLAG_STREAM_WALLS | LBM_FLAG_STREAM_PAIRS))) {
                dispatch(c->p, [&](auto v) {
                    using VT = decltype(v);
                    using R = typename VT::R;
                    if constexpr (VT::SEM == SEM_GPU) {
                        hipFuncAttributes at;
                        if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_stream_walls<R, VT::COLL, VT::TURB>)) != hipSuccess || at.localSizeBytes > 48)
                            walls_pay = false;
                    }
                });
            }
            c->stream_walls = walls_ok && !(p->flags & LBM_FLAG_NO_STREAM_WALLS) &&
                              (walls_pay || (p->flags & (LBM_FLAG_STREAM_WALLS | LBM_FLAG_STREAM_PAIRS)));
            // ... and two rows per wave (k_stream_pairs): twelve waves that all work in every iteration, 10 steps per launch by default (up
            // to SP_MAX_S), a launch that costs in proportion to its steps -- so no tile-kernel tails
            c->stream_pairs = c->stream_walls && (p->flags & LBM_FLAG_STREAM_PAIRS);
            if (c->stream_pairs) {
                c->tb_steps = p->tb_steps ? p->tb_steps : 10;
                c->tail_tiles = false;
            } else if (c->tb_steps > ST_MAX_S) {
                return (delete c, bail("tb_steps " + std::to_string(ST_MAX_S + 1) + " .. " + std::to_string(SP_MAX_S) + " need the streaming kernel with two rows per wave (a lone lattice, MRT_GPU semantics)"));
            }
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

extern "C" {

int lbm_abi_version(void) { return LBM_ABI_VERSION; }

int lbm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

lbm_ctx* lbm_create(const lbm_params* p, char* err, size_t errlen) {
    auto bail = [&](const std::string& m) -> lbm_ctx* {
        if (err && errlen) { std::snprintf(err, errlen, "%s", m.c_str()); }
        return nullptr;
    };
    {
        const std::string bad = validate_params(p);
        if (!bad.empty()) return bail(bad);
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return bail(std::string("no HIP device: ") + hipGetErrorString(e));
    if (p->device < 0 || p->device >= ndev) return bail("device ordinal out of range");
    if ((e = hipSetDevice(p->device)) != hipSuccess) return bail(std::string("hipSetDevice: ") + hipGetErrorString(e));

    std::string plan_err;
    lbm_ctx* c = plan_ctx(p, true, plan_err);
    if (!c) return bail(plan_err);
    const size_t bytes = c->lat_bytes;
    auto cleanup = [&](const std::string& m) -> lbm_ctx* { lbm_destroy(c); return bail(m); };
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) c->ncu = prop.multiProcessorCount;
    }
    // beside the streaming kernel one frame workgroup fits on a CU: as many workgroups as CUs, not more (8192^2 fp64: 187 GLUPS
    // with 64 cells = 512 workgroups, 201 with 128 = 256; profiles/r02_logs/stream_ab20.log)
    if (c->frame_beside && !p->frame_seg) {
        const long long per = 2LL * p->nx + 2LL * ((p->ny_local_min ? p->ny_local_min : p->ny_local) - 2 * c->tb_f);
        c->frame_seg = std::max(64, (int)(((per + c->ncu - 1) / c->ncu + 7) / 8 * 8));
    }
    if ((e = hipStreamCreateWithFlags(&c->s_compute, hipStreamNonBlocking)) != hipSuccess) return cleanup("hipStreamCreate");
    {   // halo exchange stream at the highest priority: its (tiny) RCCL kernels must not queue behind the
        // thousands of workgroups of the interior kernel they are meant to overlap
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (p->flags & LBM_FLAG_COMM_PRIORITY_OFF) hi = lo;   // A/B: same priority as the compute stream
        if ((e = hipStreamCreateWithPriority(&c->s_comm, hipStreamNonBlocking, hi)) != hipSuccess) return cleanup("hipStreamCreate");
    }
    if ((e = hipEventCreateWithFlags(&c->ev_edges, hipEventDisableTiming)) != hipSuccess) return cleanup("hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming)) != hipSuccess) return cleanup("hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_int, hipEventDisableTiming)) != hipSuccess) return cleanup("hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_go, hipEventDisableTiming)) != hipSuccess) return cleanup("hipEventCreate");
    if ((e = hipEventCreate(&c->ev_t0)) != hipSuccess) return cleanup("hipEventCreate");
    if ((e = hipEventCreate(&c->ev_t1)) != hipSuccess) return cleanup("hipEventCreate");
    // the two lattices (+ ftemp of the push scheme); the scratch lattices of the frame passes come on first use (ensure_scratch)
    for (int i = 0; i < (c->push ? 3 : 2); ++i) {
        if ((e = hipMalloc(&c->lat[i], bytes)) != hipSuccess) return cleanup(std::string("hipMalloc(lattice): ") + hipGetErrorString(e));
        // on the compute stream: the streams are non-blocking, a null-stream memset would race with the kernels
        if ((e = hipMemsetAsync(c->lat[i], 0, bytes, c->s_compute)) != hipSuccess) return cleanup(std::string("hipMemset: ") + hipGetErrorString(e));
    }
    if (c->batch > 1) {   // every lattice starts with the rates of lbm_params; lbm_set_relaxation() changes them one by one
        const size_t rb = c->es == 4 ? sizeof(Relax<float>) : sizeof(Relax<double>);
        if ((e = hipMalloc(&c->relax_dev, rb * c->batch)) != hipSuccess) return cleanup(std::string("hipMalloc(relaxation): ") + hipGetErrorString(e));
        for (int i = 0; i < c->batch; ++i)
            if (lbm_set_relaxation(c, i, p->omega, p->omegam, p->omega_e, p->omega_eps, p->omega_q) != LBM_OK) return cleanup(c->err);
    }
    if (lbm_init_equilibrium(c) != LBM_OK) return cleanup(c->err);
    if (c->stream && c->tail_tiles && warm_stream(c) != LBM_OK) return cleanup("warm-up launch of the streaming kernel");
    return c;
}

void lbm_destroy(lbm_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    if (c->s_compute) (void)hipStreamSynchronize(c->s_compute);
    if (c->s_comm) (void)hipStreamSynchronize(c->s_comm);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    for (int i = 0; i < NLAT; ++i)
        if (c->lat[i]) (void)hipFree(c->lat[i]);
    if (c->red_dev) (void)hipFree(c->red_dev);
    if (c->stage) (void)hipFree(c->stage);
    if (c->relax_dev) (void)hipFree(c->relax_dev);
    if (c->ev_edges) (void)hipEventDestroy(c->ev_edges);
    if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
    if (c->ev_int) (void)hipEventDestroy(c->ev_int);
    if (c->ev_go) (void)hipEventDestroy(c->ev_go);
    if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
    if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
    if (c->s_compute) (void)hipStreamDestroy(c->s_compute);
    if (c->s_comm) (void)hipStreamDestroy(c->s_comm);
    delete c;
}

const char* lbm_last_error(const lbm_ctx* c) { return c ? c->err.c_str() : "null context"; }

int lbm_init_equilibrium(lbm_ctx* c) {
    if (!c) return LBM_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->p.device));
    int rc = sync_all(c);
    if (rc) return rc;
    c->cur = 0; c->raw[0] = 1; c->raw[1] = 1; c->nsteps = 0; c->lag = 0; c->lag_valid = false; c->thin_valid = false;
    const dim3 g = grid_rows(c, c->geo.ny);
    if (c->p.dtype == LBM_F32)
        hipLaunchKernelGGL((k_init<float>), g, dim3(BLK), 0, c->s_compute, (float*)c->lat[0], c->geo, (float)c->p.uLB, c->p.turb, c->bstride);
    else
        hipLaunchKernelGGL((k_init<double>), g, dim3(BLK), 0, c->s_compute, (double*)c->lat[0], c->geo, (double)c->p.uLB, c->p.turb, c->bstride);
    HIP_TRY(c, hipGetLastError());
    return push_reset(c);
}

int lbm_set_state(lbm_ctx* c, const void* fin_host, int host_dtype) {
    if (!c || !fin_host || (host_dtype != LBM_F32 && host_dtype != LBM_F64)) return fail(c, LBM_ERR_INVALID, "lbm_set_state: bad argument");
    HIP_TRY(c, hipSetDevice(c->p.device));
    int rc = sync_all(c);
    if (rc) return rc;
    rc = ensure_stage(c, (size_t)12 * c->geo.nx * c->geo.ny * c->es * c->batch);
    if (rc) return rc;
    rc = host_to_stage(c, fin_host, host_dtype, Q * c->batch);   // [B][9][nx][ny] is B * 9 planes
    if (rc) return rc;
    c->cur = 0; c->raw[0] = 1; c->raw[1] = 1; c->nsteps = 0; c->lag = 0; c->lag_valid = false; c->thin_valid = false;
    if (c->p.dtype == LBM_F32)
        hipLaunchKernelGGL((k_import<float>), grid_tiles<float>(c), dim3(BLK), 0, c->s_compute, (const float*)c->stage, (float*)c->lat[0], c->geo, (float)c->p.uLB, c->p.turb, c->bstride);
    else
        hipLaunchKernelGGL((k_import<double>), grid_tiles<double>(c), dim3(BLK), 0, c->s_compute, (const double*)c->stage, (double*)c->lat[0], c->geo, (double)c->p.uLB, c->p.turb, c->bstride);
    HIP_TRY(c, hipGetLastError());
    rc = push_reset(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->s_compute));
    return LBM_OK;
}

int lbm_set_relaxation(lbm_ctx* c, int index, double omega, double omegam, double omega_e, double omega_eps, double omega_q) {
    if (!c || index < 0 || index >= c->batch) return fail(c, LBM_ERR_INVALID, "lbm_set_relaxation: index out of range");
    HIP_TRY(c, hipSetDevice(c->p.device));
    lbm_params q = c->p;
    q.omega = omega; q.omegam = omegam; q.omega_e = omega_e; q.omega_eps = omega_eps; q.omega_q = omega_q;
    if (c->batch == 1) {   // rates travel by value with every launch
        c->p = q;
        return LBM_OK;
    }
    // ordered behind the steps already enqueued: the copy goes through the compute stream (pageable source, so the call
    // returns only after the runtime has staged it)
    HIP_TRY(c, hipStreamSynchronize(c->s_comm));
    if (c->es == 4) {
        const Relax<float> w = relax_of<float>(q);
        HIP_TRY(c, hipMemcpyAsync((Relax<float>*)c->relax_dev + index, &w, sizeof(w), hipMemcpyHostToDevice, c->s_compute));
    } else {
        const Relax<double> w = relax_of<double>(q);
        HIP_TRY(c, hipMemcpyAsync((Relax<double>*)c->relax_dev + index, &w, sizeof(w), hipMemcpyHostToDevice, c->s_compute));
    }
    HIP_TRY(c, hipStreamSynchronize(c->s_compute));
    return LBM_OK;
}

int lbm_step(lbm_ctx* c, int nsteps) {
    if (!c || nsteps < 0) return fail(c, LBM_ERR_INVALID, "lbm_step: bad argument");
    HIP_TRY(c, hipSetDevice(c->p.device));
    return step_many(c, nsteps);
}

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
        const long long rows = c->geo.ny - (c->stream_walls ? 0 : 2 * c->tb_f);
        // (with the walls inside the first / last segment has no lead rows beyond the wall)
        wave_updates = (long long)pl.nstrips * (rows + ((long long)pl.nsegy * 2 - (c->stream_walls ? 2 : 0)) * (S - 1)) * S;
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
                                kern, S, c->use_tb ? (c->stream_walls ? 0 : c->tb_f) : 0, c->stream ? 1 : 0, c->use_vec ? 1 : 0, c->use_nt ? 1 : 0, c->deep_halo ? 1 : 0,
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

int lbm_sync(lbm_ctx* c) {
    if (!c) return LBM_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->p.device));
    return sync_all(c);
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

long long lbm_steps_done(const lbm_ctx* c) { return c ? c->nsteps : -1; }

int lbm_get_fields(lbm_ctx* c, void* u_host, void* rho_host, void* fin_host, int host_dtype) {
    if (!c || (host_dtype != LBM_F32 && host_dtype != LBM_F64)) return fail(c, LBM_ERR_INVALID, "lbm_get_fields: bad argument");
    HIP_TRY(c, hipSetDevice(c->p.device));
    int rc = sync_all(c);
    if (rc) return rc;
    const size_t n = (size_t)c->geo.nx * c->geo.ny;
    rc = ensure_stage(c, (size_t)12 * n * c->es * c->batch);
    if (rc) return rc;
    const size_t hes = host_dtype == LBM_F32 ? 4 : 8;
    if (u_host || rho_host) {
        rc = c->p.dtype == LBM_F32 ? export_macro_t<float>(c) : export_macro_t<double>(c);
        if (rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->s_compute));
        for (int b = 0; b < c->batch; ++b) {   // staging of lattice b: [ux | uy | rho]; host: u[B][2][nx][NY], rho[B][nx][NY]
            const char* st = (const char*)c->stage + (size_t)b * 3 * n * c->es;
            const size_t hn = (size_t)c->geo.nx * c->geo.NY * hes;
            if (u_host) { rc = stage_to_host(c, st, (char*)u_host + (size_t)b * 2 * hn, host_dtype, 2); if (rc) return rc; }
            if (rho_host) { rc = stage_to_host(c, st + 2 * n * c->es, (char*)rho_host + (size_t)b * hn, host_dtype, 1); if (rc) return rc; }
        }
    }
    if (fin_host) {
        rc = c->p.dtype == LBM_F32 ? export_fin_t<float>(c) : export_fin_t<double>(c);
        if (rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->s_compute));
        rc = stage_to_host(c, c->stage, fin_host, host_dtype, Q * c->batch);
        if (rc) return rc;
    }
    return LBM_OK;
}

int lbm_mean_u(lbm_ctx* c, double* mean_out) {
    if (!c || !mean_out) return fail(c, LBM_ERR_INVALID, "lbm_mean_u: bad argument");
    HIP_TRY(c, hipSetDevice(c->p.device));
    int rc = sync_all(c);
    if (rc) return rc;
    if (!c->red_dev) {
        hipError_t e = hipMalloc((void**)&c->red_dev, ((size_t)RED_BLOCKS + 1) * c->batch * sizeof(double));
        if (e != hipSuccess) return fail(c, LBM_ERR_NOMEM, std::string("hipMalloc(reduction): ") + hipGetErrorString(e));
    }
    rc = c->p.dtype == LBM_F32 ? reduce_u_t<float>(c) : reduce_u_t<double>(c);
    if (rc) return rc;
    double* res = c->red_dev + (size_t)RED_BLOCKS * c->batch;
    const double scale = 1.0 / (2.0 * (double)c->geo.nx * (double)c->geo.ny);
    hipLaunchKernelGGL(k_reduce_final, dim3((c->batch + BLK - 1) / BLK), dim3(BLK), 0, c->s_compute, c->red_dev, RED_BLOCKS, c->batch, scale, res);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(mean_out, res, (size_t)c->batch * sizeof(double), hipMemcpyDeviceToHost, c->s_compute));
    HIP_TRY(c, hipStreamSynchronize(c->s_compute));
    return LBM_OK;
}

int lbm_get_tau(lbm_ctx* c, void* tau_host, int host_dtype) {
    if (!c || !tau_host || (host_dtype != LBM_F32 && host_dtype != LBM_F64)) return fail(c, LBM_ERR_INVALID, "lbm_get_tau: bad argument");
    HIP_TRY(c, hipSetDevice(c->p.device));
    int rc = sync_all(c);
    if (rc) return rc;
    const size_t n = (size_t)c->geo.nx * c->geo.ny;
    rc = ensure_stage(c, (size_t)12 * n * c->es * c->batch);
    if (rc) return rc;
    rc = c->p.dtype == LBM_F32 ? export_tau_t<float>(c) : export_tau_t<double>(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->s_compute));
    return stage_to_host(c, c->stage, tau_host, host_dtype, c->batch);
}

int lbm_halo_elems(const lbm_ctx* c) { return c ? 3 * c->geo.nx : 0; }

int lbm_halo_export(lbm_ctx* c, int side, void* buf) {
    if (!c || !buf || (side != LBM_SIDE_LOW && side != LBM_SIDE_HIGH)) return fail(c, LBM_ERR_INVALID, "lbm_halo_export: bad argument");
    if (c->batch > 1) return fail(c, LBM_ERR_STATE, "a batch of lattices has no slab halos");
    HIP_TRY(c, hipSetDevice(c->p.device));
    const int* pl = side_planes(side);
    const int row = side == LBM_SIDE_LOW ? 0 : c->geo.ny - 1;
    const size_t rb = (size_t)c->geo.nx * c->es;
    for (int j = 0; j < 3; ++j)
        HIP_TRY(c, hipMemcpyAsync((char*)buf + j * rb, plane_row(c, c->cur, pl[j], row), rb, hipMemcpyDefault, c->s_compute));
    HIP_TRY(c, hipStreamSynchronize(c->s_compute));
    return LBM_OK;
}

int lbm_halo_import(lbm_ctx* c, int side, const void* buf) {
    if (!c || !buf || (side != LBM_SIDE_LOW && side != LBM_SIDE_HIGH)) return fail(c, LBM_ERR_INVALID, "lbm_halo_import: bad argument");
    if (c->batch > 1) return fail(c, LBM_ERR_STATE, "a batch of lattices has no slab halos");
    HIP_TRY(c, hipSetDevice(c->p.device));
    const int* pl = side_planes(side ^ 1);  // what arrives through `side` left the neighbour's opposite side
    const int row = side == LBM_SIDE_LOW ? -1 : c->geo.ny;
    const size_t rb = (size_t)c->geo.nx * c->es;
    for (int j = 0; j < 3; ++j) {
        int lo, hi;
        halo_range(c, pl[j], &lo, &hi);
        HIP_TRY(c, hipMemcpyAsync(plane_row(c, c->cur, pl[j], row) + (size_t)lo * c->es, (const char*)buf + j * rb + (size_t)lo * c->es,
                                  (size_t)(hi - lo + 1) * c->es, hipMemcpyDefault, c->s_compute));
    }
    HIP_TRY(c, hipStreamSynchronize(c->s_compute));
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

long long lbm_halo_rows_elems(const lbm_ctx* c, int nrows) {
    if (!c || nrows < 1 || nrows >= GHY) return 0;
    return (long long)nrows * (c->p.turb ? Q + 2 : Q) * c->geo.pitch;
}

namespace {
int copy_rows(lbm_ctx* c, int side, int nrows, void* buf, bool out) {
    if (!c || !buf || (side != LBM_SIDE_LOW && side != LBM_SIDE_HIGH) || nrows < 1 || nrows >= GHY || nrows > c->geo.ny)
        return fail(c, LBM_ERR_INVALID, "lbm_halo_export_rows / lbm_halo_import_rows: bad argument (1 <= nrows <= " + std::to_string(GHY - 1) +
                                        ", the ghost rows of a lattice, and <= ny_local)");
    if (c->batch > 1) return fail(c, LBM_ERR_STATE, "a batch of lattices has no slab halos");
    HIP_TRY(c, hipSetDevice(c->p.device));
    const RowBlocks b = deep_blocks(c, c->cur, out ? deep_send_row0(c, side, nrows) : deep_recv_row0(c, side, nrows), nrows);
    const size_t bytes = b.elems * c->es;
    for (int i = 0; i < b.n; ++i) {
        char* p = (char*)buf + (size_t)i * bytes;
        HIP_TRY(c, hipMemcpyAsync(out ? (void*)p : (void*)b.ptr[i], out ? (const void*)b.ptr[i] : (const void*)p, bytes, hipMemcpyDefault, c->s_compute));
    }
    HIP_TRY(c, hipStreamSynchronize(c->s_compute));
    return LBM_OK;
}
}  // namespace

int lbm_halo_export_rows(lbm_ctx* c, int side, int nrows, void* buf) { return copy_rows(c, side, nrows, buf, true); }
int lbm_halo_import_rows(lbm_ctx* c, int side, int nrows, const void* buf) { return copy_rows(c, side, nrows, const_cast<void*>(buf), false); }

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

int lbm_comm_unique_id(void* uid_out128) {
    if (!uid_out128) return LBM_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    if (!rccl().ok) return LBM_ERR_COMM;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return LBM_ERR_COMM;
    std::memcpy(uid_out128, &id, sizeof(id));
    return LBM_OK;
}

int lbm_comm_init(lbm_ctx* c, int nranks, int rank, const void* uid128) {
    if (!c || !uid128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, LBM_ERR_INVALID, "lbm_comm_init: bad argument");
    if (c->comm) return fail(c, LBM_ERR_STATE, "communicator already attached");
    if (c->batch > 1 || c->push) return fail(c, LBM_ERR_STATE, "a batch of lattices / kernel = PUSH cannot be slab-decomposed");
    // rank r holds the r-th slab from the lid: the exchange partners are rank - 1 / rank + 1
    if ((rank > 0) != has_neighbour(c, LBM_SIDE_LOW) || (rank < nranks - 1) != has_neighbour(c, LBM_SIDE_HIGH))
        return fail(c, LBM_ERR_INVALID, "lbm_comm_init: rank 0 must hold the slab at the lid (y0 = 0), the last rank the one at the bottom wall, "
                                        "every other rank a slab in between");
    if (!rccl().ok) return fail(c, LBM_ERR_COMM, rccl().err);
    HIP_TRY(c, hipSetDevice(c->p.device));
    ncclUniqueId id;
    std::memcpy(&id, uid128, sizeof(id));
    NCCL_TRY(c, rccl().CommInitRank(&c->comm, nranks, id, rank));
    c->nranks = nranks;
    c->rank = rank;
    c->thin_valid = false;
    if (nranks > 1) {
        // Neighbours must run the same launch plan (they post matching send / receive sequences): compare it once.
        constexpr int NW = 16;
        const int32_t mine[NW] = {LBM_ABI_VERSION, c->p.nx, c->p.ny, c->p.dtype, c->p.semantics, c->p.turb, c->geo.pitch,
                                  c->geo.row != c->geo.pitch ? 1 : 0, c->use_tb ? (c->stream ? 2 : 1) : 0, c->tb_steps, c->tb_f, c->deep_halo ? 1 : 0,
                                  c->frame_fused ? 1 : 0, c->lazy_lag ? 1 : 0, c->p.collision, c->p.arith};
        // (UNEXECUTED ON HARDWARE until a run with two GPUs exists: every box so far had one.)  The three blocks [mine | from LOW | from
        // HIGH] are built on the host and uploaded by ONE synchronous copy, so nothing on the null stream can race with the receives
        // that s_comm (a non-blocking stream) enqueues below; a failed send / receive still closes the RCCL group, and every failure
        // path gives the communicator back.
        auto drop_comm = [&] { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; c->nranks = 1; c->rank = 0; };
        int32_t host[3 * NW];
        std::memcpy(host, mine, sizeof(mine));
        std::memset(host + NW, 0xff, 2 * NW * sizeof(int32_t));
        int32_t* dev = nullptr;
        hipError_t e = hipMalloc((void**)&dev, sizeof(host));
        if (e == hipSuccess) e = hipMemcpy(dev, host, sizeof(host), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            if (dev) (void)hipFree(dev);
            drop_comm();
            return fail(c, LBM_ERR_HIP, std::string("lbm_comm_init (plan check): ") + hipGetErrorString(e));
        }
        ncclResult_t r = rccl().GroupStart();
        if (r == ncclSuccess) {
            for (int side = 0; side < 2 && r == ncclSuccess; ++side) {
                if (!has_neighbour(c, side)) continue;
                const int peer = side == LBM_SIDE_LOW ? rank - 1 : rank + 1;
                r = rccl().Send(dev, NW, ncclInt32, peer, c->comm, c->s_comm);
                if (r == ncclSuccess) r = rccl().Recv(dev + (1 + side) * NW, NW, ncclInt32, peer, c->comm, c->s_comm);
            }
            const ncclResult_t r_end = rccl().GroupEnd();   // (always: a group left open would swallow every later RCCL call)
            if (r == ncclSuccess) r = r_end;
        }
        int32_t theirs[2 * NW];
        if (r == ncclSuccess) {
            e = hipStreamSynchronize(c->s_comm);
            if (e == hipSuccess) e = hipMemcpy(theirs, dev + NW, sizeof(theirs), hipMemcpyDeviceToHost);
        }
        (void)hipFree(dev);
        if (r != ncclSuccess) { const std::string m = rccl().GetErrorString(r); drop_comm(); return fail(c, LBM_ERR_COMM, "lbm_comm_init (plan check): " + m); }
        if (e != hipSuccess) { drop_comm(); return fail(c, LBM_ERR_HIP, std::string("lbm_comm_init (plan check): ") + hipGetErrorString(e)); }
        static const char* what[NW] = {"ABI version", "nx", "ny", "dtype", "semantics", "turb", "row pitch", "layout", "steps-per-launch path (0 none, 1 tile, 2 streaming kernel)",
                                       "steps per launch", "frame width", "deep halo", "fused frame", "lazy lag", "collision", "arith"};
        for (int side = 0; side < 2; ++side) {
            if (!has_neighbour(c, side)) continue;
            for (int i = 0; i < NW; ++i)
                if (theirs[side * NW + i] != mine[i]) {
                    drop_comm();
                    return fail(c, LBM_ERR_STATE, std::string("lbm_comm_init: rank ") + std::to_string(side == LBM_SIDE_LOW ? rank - 1 : rank + 1) +
                                                  " runs a different launch plan (" + what[i] + ": " + std::to_string(theirs[side * NW + i]) + " there, " +
                                                  std::to_string(mine[i]) + " here); create every rank with the same parameters and "
                                                  "lbm_params.ny_local_min = the smallest slab");
                }
        }
    }
    return LBM_OK;
}

int lbm_comm_loopback(lbm_ctx* c) {
    if (!c) return LBM_ERR_INVALID;
    if (c->comm) return fail(c, LBM_ERR_STATE, "communicator already attached");
    if (c->geo.y0 == 0 || c->geo.y0 + c->geo.ny == c->geo.NY)
        return fail(c, LBM_ERR_INVALID, "lbm_comm_loopback needs a slab that touches neither the lid nor the bottom wall");
    if (!rccl().ok) return fail(c, LBM_ERR_COMM, rccl().err);
    HIP_TRY(c, hipSetDevice(c->p.device));
    ncclUniqueId id;
    NCCL_TRY(c, rccl().GetUniqueId(&id));
    NCCL_TRY(c, rccl().CommInitRank(&c->comm, 1, id, 0));
    c->nranks = 1;
    c->rank = 0;
    c->loopback = true;
    c->thin_valid = false;
    return LBM_OK;
}

int lbm_copy_bandwidth(lbm_ctx* c, size_t bytes, int iters, double* gbps) {
    if (!c || !gbps || iters < 1 || bytes < 16) return fail(c, LBM_ERR_INVALID, "lbm_copy_bandwidth: bad argument");
    HIP_TRY(c, hipSetDevice(c->p.device));
    bytes &= ~(size_t)15;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) {
        if (a) (void)hipFree(a);
        return fail(c, LBM_ERR_NOMEM, "lbm_copy_bandwidth: hipMalloc failed");
    }
    (void)hipMemsetAsync(a, 1, bytes, c->s_compute);
    const size_t n = bytes / 16;
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(BLK), 0, c->s_compute, (const uint4*)a, (uint4*)b, n);
    (void)hipEventRecord(c->ev_t0, c->s_compute);
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(BLK), 0, c->s_compute, (const uint4*)a, (uint4*)b, n);
    (void)hipEventRecord(c->ev_t1, c->s_compute);
    hipError_t e = hipEventSynchronize(c->ev_t1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1);
    (void)hipFree(a);
    (void)hipFree(b);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, std::string("lbm_copy_bandwidth: ") + hipGetErrorString(e));
    *gbps = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;
    return LBM_OK;
}

int lbm_fma_rate(lbm_ctx* c, double ms_total, double* tflops) {
    if (!c || !tflops || !(ms_total > 0) || ms_total > 2000) return fail(c, LBM_ERR_INVALID, "lbm_fma_rate: bad argument (0 < ms <= 2000)");
    HIP_TRY(c, hipSetDevice(c->p.device));
    float* sink = nullptr;
    const int blocks = c->ncu * 8, iters = 4096;
    HIP_TRY(c, hipMalloc((void**)&sink, (size_t)blocks * BLK * sizeof(float)));
    auto run = [&](int n, float* ms) -> hipError_t {
        (void)hipEventRecord(c->ev_t0, c->s_compute);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_fma_burn, dim3(blocks), dim3(BLK), 0, c->s_compute, sink, iters, 1.0f + (float)i);
        (void)hipEventRecord(c->ev_t1, c->s_compute);
        hipError_t e = hipEventSynchronize(c->ev_t1);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, c->ev_t0, c->ev_t1);
        return e;
    };
    float ms1 = 0.f, ms = 0.f;
    hipError_t e = run(1, &ms1);                                    // calibration (also the first, slower, launch)
    const int n = e == hipSuccess && ms1 > 0 ? std::max(1, (int)(ms_total / ms1)) : 1;
    if (e == hipSuccess) e = run(n, &ms);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(c, LBM_ERR_HIP, std::string("lbm_fma_rate: ") + hipGetErrorString(e));
    // per thread and iteration: 16 packed FMAs = 32 FMAs = 64 flop
    *tflops = (double)n * blocks * BLK * (double)iters * 64.0 / (ms * 1e-3) / 1e12;
    return LBM_OK;
}

}  // extern "C"
