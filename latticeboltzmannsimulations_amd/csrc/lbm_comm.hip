// lbm_comm.hip -- the lazily bound RCCL table, ghost-row exchanges between y-slabs (one step and S steps deep), their ordering
// against the compute stream, and the host-transported halo entry points.
#include "lbm_host.hpp"

namespace lbmhost {

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

#ifdef LBM_DEBUG
static bool debug_skip_exchange() {   // timing diagnostic of debug builds only: results between slabs are wrong
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
    if (c->edge_rows < rows) {
        const int rc = flush_int(c);
        if (rc) return rc;
        HIP_TRY(c, hipStreamWaitEvent(c->s_comm, c->ev_int, 0));
    }
    return LBM_OK;
}

// later single-stream work (export, timing event, externally driven calls) must see the s_comm results
int join_comm(lbm_ctx* c) {
    HIP_TRY(c, hipEventRecord(c->ev_halo, c->s_comm));
    HIP_TRY(c, hipStreamWaitEvent(c->s_compute, c->ev_halo, 0));
    return LBM_OK;
}
// lbm_halo_export_rows / lbm_halo_import_rows: the send / receive blocks of an S-step exchange, through host memory.
static int copy_rows(lbm_ctx* c, int side, int nrows, void* buf, bool out) {
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
}  // namespace lbmhost

using namespace lbmhost;

extern "C" {

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

long long lbm_halo_rows_elems(const lbm_ctx* c, int nrows) {
    if (!c || nrows < 1 || nrows >= GHY) return 0;
    return (long long)nrows * (c->p.turb ? Q + 2 : Q) * c->geo.pitch;
}

int lbm_halo_export_rows(lbm_ctx* c, int side, int nrows, void* buf) { return copy_rows(c, side, nrows, buf, true); }
int lbm_halo_import_rows(lbm_ctx* c, int side, int nrows, const void* buf) { return copy_rows(c, side, nrows, const_cast<void*>(buf), false); }

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
}  // extern "C"
