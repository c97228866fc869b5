// lbm_hip.hip -- liblbm_hip.so: context life cycle, state upload, field export and the bandwidth / FMA probes of the C ABI declared
// in include/lbm.h (the other host units: lbm_plan.hip, lbm_launch.hip, lbm_comm.hip; shared declarations: lbm_host.hpp).
// gfx950 only.  See DESIGN.md for the data layout and the per-kernel roofline notes.
#include "lbm_host.hpp"

namespace lbmhost {

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

int ensure_stage(lbm_ctx* c, size_t bytes) {
    if (c->stage_bytes >= bytes) return LBM_OK;
    if (c->stage) { (void)hipFree(c->stage); c->stage = nullptr; c->stage_bytes = 0; }
    hipError_t e = hipMalloc(&c->stage, bytes);
    if (e != hipSuccess) return fail(c, LBM_ERR_NOMEM, std::string("hipMalloc(staging): ") + hipGetErrorString(e));
    c->stage_bytes = bytes;
    return LBM_OK;
}
int sync_all(lbm_ctx* c) {
    if (!spin_until_ready([&] { return hipStreamQuery(c->s_compute); })) HIP_TRY(c, hipStreamSynchronize(c->s_compute));
    if (!spin_until_ready([&] { return hipStreamQuery(c->s_comm); })) HIP_TRY(c, hipStreamSynchronize(c->s_comm));
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
}  // namespace lbmhost

using namespace lbmhost;

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
    // (the events between the two streams of this context order kernels of ONE device: no system-scope release with every record -- a
    // slab's unit records two and waits for two; 4096 x 512 slab in loopback 189 -> 195 GLUPS, 4096 x 1024 246 -> 251)
    constexpr unsigned LBM_EVENT_FENCE = hipEventDisableSystemFence;
    if ((e = hipEventCreateWithFlags(&c->ev_edges, hipEventDisableTiming | LBM_EVENT_FENCE)) != hipSuccess) return cleanup("hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming)) != hipSuccess) return cleanup("hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_int, hipEventDisableTiming | LBM_EVENT_FENCE)) != hipSuccess) return cleanup("hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_go, hipEventDisableTiming | LBM_EVENT_FENCE)) != hipSuccess) return cleanup("hipEventCreate");
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

int lbm_sync(lbm_ctx* c) {
    if (!c) return LBM_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->p.device));
    return sync_all(c);
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
