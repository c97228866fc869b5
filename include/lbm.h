/* lbm.h -- C ABI of liblbm_hip.so: MI355X-native D2Q9 lid-driven-cavity lattice-Boltzmann hot path.
 *
 * Drop-in boundary for the PyCUDA usage of the reference's GPU script.  Each entry point
 * cites the reference interface it replaces (paths relative to the reference repo root).
 * Plain C types only: no torch / C++ types cross this boundary.  Every function returning
 * int returns LBM_OK (0) or a negative lbm_status; the text is available from
 * lbm_last_error().  No C++ exception crosses the ABI.  One host thread per context.
 *
 * Host array convention (same as the reference scripts): fin[9][X][Y], u[2][X][Y],
 * rho[X][Y], C order, y fastest, y = 0 is the moving lid (MRT.py:192-209,
 * MRT_GPU.py:207-229).  On the device the library keeps one padded plane per direction,
 * x fastest (the transposition the reference does on the host at MRT_GPU.py:283-289 and
 * MRT_GPU.py:758-760 is done inside lbm_set_state / lbm_get_fields).
 */
#ifndef LBM_H
#define LBM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_ABI_VERSION 3

typedef enum lbm_status {
    LBM_OK = 0,
    LBM_ERR_INVALID = -1, /* bad argument / unsupported combination */
    LBM_ERR_HIP = -2,     /* a HIP runtime call failed (text in lbm_last_error) */
    LBM_ERR_NOMEM = -3,
    LBM_ERR_STATE = -4,   /* call not valid in the current context state */
    LBM_ERR_COMM = -5     /* RCCL failure */
} lbm_status;

enum { LBM_F32 = 0, LBM_F64 = 1 };                      /* storage and arithmetic type */
enum { LBM_SRT = 0, LBM_TRT = 1, LBM_MRT = 2 };         /* RT = 'SRT' | 'TRT' | 'MRT'  (MRT_GPU.py:48) */
enum { LBM_SEM_MRT_PY = 0, LBM_SEM_MRT_GPU = 1 };       /* streaming windows + wall rules of MRT.py:404-453
                                                           or of MRT_GPU.py:412,674-692 */
enum { LBM_KERNEL_AUTO = 0,      /* fastest applicable: STREAM (large lattices), TB (lattices from 64 x 64 cells), else VEC, else GENERIC */
       LBM_KERNEL_GENERIC = 1,   /* one step per launch, one thread per cell (all semantics) */
       LBM_KERNEL_VEC = 2,       /* one step per launch, 16 B per access (MRT_GPU semantics) */
       LBM_KERNEL_TB = 3,        /* several (3 .. 5) time steps per launch: tiles through LDS + the wall frame */
       LBM_KERNEL_PUSH = 4,      /* the reference's own scheme for A/B: collide-and-push into a persistent second array, then a
                                    wall-rule + copy kernel (funRT + funBC, MRT_GPU.py:339-698); one whole lattice, no closure */
       LBM_KERNEL_STREAM = 5 };  /* up to 8 time steps per launch, streamed down column strips (rows held in registers, neighbour rows
                                    through LDS): what AUTO picks for large lone lattices; otherwise as TB */
enum { LBM_LAYOUT_AUTO = 0, LBM_LAYOUT_PLANES = 1, LBM_LAYOUT_ROWS = 2 }; /* device arrays: [k][y][x] or [y][k][x] */
enum { LBM_ARITH_STRICT = 0, LBM_ARITH_FAST = 1 };
enum { LBM_SIDE_LOW = 0, LBM_SIDE_HIGH = 1 };           /* slab neighbour towards smaller / larger y */
/* lbm_params.flags: A/B switches of the launch plan (all off = the measured defaults); results never depend on them */
enum { LBM_FLAG_NO_DEEP_HALO = 1,        /* between slabs: a one-row exchange after every frame pass of a multi-step launch instead
                                            of ONE exchange of S complete rows per launch */
       LBM_FLAG_FRAME_UNFUSED = 2,       /* one launch per frame pass instead of all passes inside the tile launch */
       LBM_FLAG_FRAME_FUSED_BATCH = 4,   /* (r01: batches too run the frame passes inside the tile launch; the default since r02, accepted) */
       LBM_FLAG_NO_FRAME_LDS = 8,        /* intermediate frame passes through scratch lattices instead of LDS windows */
       LBM_FLAG_NT_ON = 16,              /* non-temporal loads / stores on (default: lattices above 192 MiB) ... */
       LBM_FLAG_NT_OFF = 32,             /* ... or off */
       LBM_FLAG_COMM_PRIORITY_OFF = 64,  /* communication stream at the compute stream's priority */
       LBM_FLAG_EAGER_LAG = 128,         /* every lbm_step() call ends with a single step (so that the lattice of the step before
                                            the last exists) instead of recomputing it when lbm_get_fields asks for u / rho */
       LBM_FLAG_FRAME_BESIDE_ON = 512,   /* kernel STREAM, lone lattice: the wall frame as a kernel of its own on the second stream, beside the */
       LBM_FLAG_FRAME_BESIDE_OFF = 1024, /* streaming workgroups, always / never (default: when the streaming kernel variant leaves registers free) */
       LBM_FLAG_FRAME_NARROW = 2048,     /* frame passes that go through the scratch lattices: workgroups of 256 threads instead of 1024 */
       LBM_FLAG_NO_EDGE_FIRST = 4096,    /* streaming kernel between slabs: do not hold the bulk launch back behind the edge launch */
       LBM_FLAG_NO_EDGE_RESERVE = 8192,  /* ... and do not plan a one-round bulk launch on fewer CUs to leave some to the edge workgroups */
       LBM_FLAG_NO_XCD_BANDS = 16384,    /* streaming kernel: workgroup i takes segment i (default: every XCD a contiguous run of segments) */
       LBM_FLAG_NO_TAIL_TILES = 32768,   /* streaming contexts with a wall frame: units of 3 .. 5 steps through the streaming kernel too (default:
                                            the tile kernel; with the walls inside the tails stay on the streaming kernel anyway) */
       LBM_FLAG_STREAM_WALLS = 65536,    /* kernel STREAM, MRT_GPU semantics, a lone lattice or a slab with the deep halo: the cells next to the walls */
       LBM_FLAG_NO_STREAM_WALLS = 262144,/* inside the streaming kernel (k_stream_walls / k_stream_walls_slab: no frame), always / never (default: for the
                                            operator variants whose kernel keeps its level loop free of scratch traffic -- MRT, SRT, no closure) */
       LBM_FLAG_STREAM_PAIRS = 131072 }; /* ... the walls inside AND two rows per wave, twelve waves, at most 10 steps per launch (k_stream_pairs:
                                            an r03 experiment, no faster than k_stream_walls -- DESIGN 2.4; kept for A/B) */

/* The knobs of the reference script (MRT_GPU.py:38-93) as run-time parameters.  The
 * reference bakes them into the CUDA source by '%'-formatting (MRT_GPU.py:422,531,662) and
 * recompiles per parameter set; here the code object is compiled ahead of time for gfx950. */
typedef struct lbm_params {
    int32_t struct_size; /* = sizeof(lbm_params) */
    int32_t nx;          /* xsize */
    int32_t ny;          /* ysize of the WHOLE lattice */
    int32_t y0;          /* first global row owned by this context (0 when not slab-decomposed) */
    int32_t ny_local;    /* rows owned by this context (= ny when not slab-decomposed, >= 2) */
    int32_t dtype;       /* LBM_F32 | LBM_F64 */
    int32_t collision;   /* LBM_SRT | LBM_TRT | LBM_MRT */
    int32_t semantics;   /* LBM_SEM_MRT_PY | LBM_SEM_MRT_GPU */
    int32_t kernel;      /* LBM_KERNEL_* */
    int32_t turb;        /* 0 | 1: Smagorinsky closure of MRT_GPU.py:368-387 (MRT_GPU semantics only) */
    int32_t device;      /* HIP device ordinal (reference: cuda.Device(0), MRT_GPU.py:29) */
    int32_t layout;      /* LBM_LAYOUT_* (device-side only; host arrays are unaffected) */
    int32_t batch;       /* 0 / 1: one lattice.  B > 1: B independent cavities of the same size and scheme advanced by
                            the same launches, each with its own relaxation rates -- the Reynolds sweep that
                            MRT_GPU_datagen.py:55-57,879-902 runs one lattice after the other.  Host arrays gain a
                            leading [B] axis.  Not combinable with slabs. */
    int32_t arith;       /* LBM_ARITH_STRICT (0, default): every operation in the reference's order, results bit-identical to
                            the CPU restatement in oracle/.  LBM_ARITH_FAST: the MRT operator in an algebraically identical
                            factored form with fused multiply-adds (about half the arithmetic); u = j * rcp(rho) and the
                            Smagorinsky closure's divisions / square root by the hardware's reciprocal / square-root
                            instructions (fp32: 1 ulp; fp64: refined by Newton steps).
                            Results agree with the strict form to rounding, not bit for bit.  MRT_GPU semantics only
                            (with LBM_SEM_MRT_PY the strict form is used). */
    int32_t ny_local_min; /* slabs: the smallest ny_local of ALL ranks (0: = ny_local).  The launch plan (steps per launch, frame
                            width, deep halo) is derived from it, so that every rank of a decomposition runs the same exchange
                            protocol whatever its own share of the rows; lbm_comm_init() cross-checks the plan with both neighbours. */
    int32_t tb_steps;    /* 0: measured default.  2 .. 5 (STREAM: 2 .. 8; a lone MRT_GPU lattice, two rows per wave: 2 .. 10): time steps per launch of the multi-step path (A/B, tests) */
    int32_t frame_seg;   /* 0: default by lattice size.  >= 8: cells of the wall frame per workgroup of the fused frame passes */
    int32_t flags;       /* LBM_FLAG_* bits, 0 = defaults */
    double uLB;          /* lid velocity, MRT_GPU.py:57 */
    double omega;        /* = omegap = omega_nu, MRT_GPU.py:65 */
    double omegam;       /* TRT, MRT_GPU.py:80 */
    double omega_e;      /* MRT, MRT_GPU.py:89 */
    double omega_eps;    /* MRT, MRT_GPU.py:90 */
    double omega_q;      /* MRT, MRT_GPU.py:90 */
} lbm_params;

typedef struct lbm_ctx lbm_ctx;

/* --- library ----------------------------------------------------------------------- */
int lbm_abi_version(void);
/* replaces: pycuda.autoinit device discovery (MRT_GPU.py:23-30) */
int lbm_device_count(void);

/* --- context ----------------------------------------------------------------------- */
/* replaces: import pycuda.autoinit + cuda.mem_alloc x8 + SourceModule/get_function
 * (MRT_GPU.py:23-30,309-316,701-703).  Returns NULL on failure with the reason in err. */
lbm_ctx* lbm_create(const lbm_params* p, char* err, size_t errlen);
/* replaces: PyCUDA garbage-collected DeviceAllocation / context teardown at exit */
void lbm_destroy(lbm_ctx* c);
const char* lbm_last_error(const lbm_ctx* c);

/* --- state in ---------------------------------------------------------------------- */
/* replaces: fin = equ(rho=1, InitVel) on the host + memcpy_htod (MRT_GPU.py:262-267,323-328);
 * computed on the device. */
int lbm_init_equilibrium(lbm_ctx* c);
/* replaces: per-plane transpose + cuda.memcpy_htod(fin_g, fin) (MRT_GPU.py:283-289,323).
 * fin_host is the WHOLE-lattice array fin[9][nx][ny]; the context reads its own rows
 * y0 .. y0+ny_local-1.  host_dtype is LBM_F32 or LBM_F64 (converted if it differs).
 * With batch = B > 1: fin[B][9][nx][ny]. */
int lbm_set_state(lbm_ctx* c, const void* fin_host, int host_dtype);

/* replaces: the per-Reynolds-number recompilation of the kernel source (MRT_GPU_datagen.py:57,63-93 -> 701-703):
 * sets the relaxation rates of lattice `index` of the batch (0 for a single lattice) for all later steps. */
int lbm_set_relaxation(lbm_ctx* c, int index, double omega, double omegam, double omega_e, double omega_eps,
                       double omega_q);

/* --- time loop --------------------------------------------------------------------- */
/* replaces: funRT(...); funBC(...) launched from the Python loop (MRT_GPU.py:707-732).
 * Enqueues nsteps fused steps and returns without waiting (errors surface at the next
 * synchronising call, like pycuda LaunchError).  Large lattices advance several steps per launch
 * (lbm_next_unit).  With an RCCL communicator attached the halo exchange with the slab neighbours is
 * part of every launch unit.  A slab (y0 > 0 or y0 + ny_local < ny) WITHOUT a communicator is refused
 * (LBM_ERR_STATE): its ghost rows would never be filled -- use the externally driven calls below. */
int lbm_step(lbm_ctx* c, int nsteps);
/* replaces: the implicit synchronisation of cuda.memcpy_dtoh (MRT_GPU.py:755) */
int lbm_sync(lbm_ctx* c);
/* replaces: the commented cuda.Event timing (MRTTiledPull.py:364-365,536-549): runs nsteps
 * steps between two HIP events on the compute stream and returns the elapsed milliseconds */
int lbm_time_steps(lbm_ctx* c, int nsteps, double* ms);
/* iterations performed since the last lbm_init_equilibrium / lbm_set_state */
long long lbm_steps_done(const lbm_ctx* c);
/* The launch plan of lbm_step(): number of time steps the next launch unit advances when `steps_left` remain
 * (1 = a single step; S > 1 = one multi-step launch of S steps).  Depends only on the parameters, ny_local_min and
 * whether the lattice has just been uploaded -- the same on every rank of a decomposition. */
int lbm_next_unit(const lbm_ctx* c, int steps_left);

/* The launch plan as text (for logs and for bench.py's roofline line): one line of `key=value` pairs -- kernel of the multi-step
 * units (k_stream | k_stepS_deep | k_step2_deep | none), steps per launch, frame width, workgroups and strip rows of a launch, ...
 * Returns the length written (truncated to len - 1), negative on a bad argument. */
int lbm_describe(const lbm_ctx* c, char* buf, size_t len);
/* Dry run, NO device needed: the text lbm_describe() would give for lbm_create(p), followed by ` units=S1,S2,...` -- the launch units
 * lbm_step(steps) would run from a freshly initialised lattice.  Derived from lbm_params alone, so a launcher can check, before any
 * rank touches a GPU, that all ranks of a decomposition plan the same kernel / steps per launch / frame / deep halo and the same
 * units (lbm_comm_init cross-checks the same items at run time).  ncu: compute units to plan for (0 = 256).  Invalid parameters:
 * returns a negative status and writes `error: <reason>`.  (No counterpart in the reference, which runs one GPU.) */
int lbm_plan(const lbm_params* p, int ncu, int steps, char* buf, size_t len);

/* --- state out --------------------------------------------------------------------- */
/* replaces: cuda.memcpy_dtoh(fin, ftemp_g); memcpy_dtoh(rho, rho_g); memcpy_dtoh(u, u_g)
 * + transposes (MRT_GPU.py:755-760).  Synchronises.  u_host[2][nx][ny], rho_host[nx][ny]
 * receive the macroscopic fields computed in the LAST iteration (one-step lag of the
 * reference: u_g/rho_g are written inside funRT before collide/stream); fin_host
 * [9][nx][ny] receives the current populations (post stream + wall rules).  Any pointer
 * may be NULL.  Whole-lattice arrays; only this context's rows are written.
 * With batch = B > 1: u_host[B][2][nx][ny], rho_host[B][nx][ny], fin_host[B][9][nx][ny].
 * (When the last launch unit advanced S > 1 steps, the lattice of the step before the last is recomputed here from the
 * unit's source lattice, S - 1 steps, bit-identically; nothing of that is on the path of lbm_step.) */
int lbm_get_fields(lbm_ctx* c, void* u_host, void* rho_host, void* fin_host, int host_dtype);
/* replaces: the download of u_g + np.mean(u) of the convergence test (MRT_GPU.py:883, MRT_GPU_datagen.py:862-871) by a
 * reduction on the device: mean_out[b] = mean over both components and all cells of this context's rows of the u that
 * lbm_get_fields would return, accumulated in double in a fixed order (deterministic); one double per lattice of the batch
 * crosses PCIe.  (The reference's criterion is defined on NumPy's float32 pairwise mean of the downloaded field; the two
 * means differ in the last bits of a float -- the host form stays available through lbm_get_fields.) */
int lbm_mean_u(lbm_ctx* c, double* mean_out);
/* replaces: taus_g (MRT_GPU.py:387; never downloaded by the reference, its dashboard prints mean(tauS), MRT_GPU.py:862-866).
 * tau_host[nx][ny] receives the relaxation time tau + tau_turbulent of the LAST iteration; with turb = 0 the constant
 * 1 / omega.  With batch = B > 1: tau_host[B][nx][ny]. */
int lbm_get_tau(lbm_ctx* c, void* tau_host, int host_dtype);

/* --- slab decomposition, externally driven exchange ---------------------------------- */
/* No reference counterpart (the reference is single-GPU, MRT_GPU.py:29).  A step of a slab
 * is split so that a host-language driver can move halos with any transport:
 *     lbm_step_edges()      rows 0 and ny_local-1 (read the ghost rows imported last)
 *     lbm_step_interior()   rows 1 .. ny_local-2
 *     lbm_step_finish()     swap lattices, count the step
 *     lbm_halo_export(side) on both neighbours -> transport -> lbm_halo_import(side)
 * i.e. every step is followed by the exchange of the rows it wrote (lbm_get_fields needs
 * current ghost rows to return the populations of the slab's first and last row).
 * lbm_halo_elems() = elements of one packed halo (3 planes x nx).  buf may be device or
 * host memory.
 *
 * Multi-step launch units between slabs (MRT_GPU semantics; what lbm_step runs when lbm_next_unit() > 1): before a unit of
 * S steps every slab hands the S COMPLETE rows next to each interface to its neighbour --
 *     lbm_halo_export_rows(side, S) -> transport -> lbm_halo_import_rows(side, S)      (lbm_halo_rows_elems(S) elements)
 * -- then lbm_step_unit(S) advances the slab S steps with no further communication: the frame passes recompute a shrinking
 * band of the neighbour's rows.  lbm_step_unit is exactly the launch sequence lbm_step uses between ranks (same kernels,
 * streams and events), minus the RCCL calls; lbm_step = this protocol with the transport inside.
 * Limits: 1 <= nrows <= 9 (a lattice carries 10 ghost rows per side; the tenth serves the recomputation of the lagged fields) and
 * nrows <= ny_local; unit_steps = 3 .. the context's steps per launch (at most 8; exactly 2 for tb_steps = 2), and lbm_next_unit
 * never plans a unit shorter than 4 on a slab. */
int lbm_halo_elems(const lbm_ctx* c);
int lbm_halo_export(lbm_ctx* c, int side, void* buf);
int lbm_halo_import(lbm_ctx* c, int side, const void* buf);
int lbm_step_edges(lbm_ctx* c);
int lbm_step_interior(lbm_ctx* c);
int lbm_step_finish(lbm_ctx* c);
long long lbm_halo_rows_elems(const lbm_ctx* c, int nrows);
int lbm_halo_export_rows(lbm_ctx* c, int side, int nrows, void* buf);
int lbm_halo_import_rows(lbm_ctx* c, int side, int nrows, const void* buf);
int lbm_step_unit(lbm_ctx* c, int unit_steps);

/* --- slab decomposition, RCCL exchange inside lbm_step -------------------------------- */
/* uid_out: 128 bytes (ncclUniqueId) created on one rank and distributed by the caller
 * (e.g. torch.distributed broadcast).  After lbm_comm_init, lbm_step() exchanges halos with
 * rank-1 / rank+1 by ncclSend/ncclRecv on a second HIP stream, overlapped with the
 * interior rows: one row of three directions before a single step; before a launch that advances S
 * steps, the S complete rows next to each interface in one message per side (MRT_GPU semantics).
 * Rank r must hold the r-th slab from the lid (rank 0: y0 = 0, last rank: y0 + ny_local = ny).  lbm_comm_init compares
 * the launch plan (steps per launch, frame width, deep halo, row pitch, planes) with both neighbours and fails with
 * LBM_ERR_STATE if they differ (pass the same lbm_params.ny_local_min on every rank).
 * RCCL is bound with dlopen on the first lbm_comm_* call and the copy already mapped in the process is the one taken.  A process
 * that also loads PyTorch must import it BEFORE that call: created a communicator first and imported torch afterwards, it aborts in
 * the exit handlers (the Python host takes care of the order itself, solver._one_rccl). */
int lbm_comm_unique_id(void* uid_out128);
int lbm_comm_init(lbm_ctx* c, int nranks, int rank, const void* uid128);
/* Diagnostic for one-GPU machines: attaches a ONE-rank RCCL communicator and makes the slab its own
 * neighbour on both sides (its last row arrives in its top ghost row and vice versa, i.e. periodic in y),
 * so that lbm_step() runs the complete exchange path -- edge/interior split, second stream, events,
 * ncclSend/ncclRecv from and into lattice memory.  Only for a slab with y0 > 0 and y0 + ny_local < ny. */
int lbm_comm_loopback(lbm_ctx* c);

/* --- measurement --------------------------------------------------------------------- */
/* Device-to-device streaming copy of `bytes` bytes (read + write), `iters` times; returns
 * achieved (read+write) GB/s: the achievable-bandwidth denominator next to the 8 TB/s peak. */
int lbm_copy_bandwidth(lbm_ctx* c, size_t bytes, int iters, double* gbps);
/* About `ms` milliseconds of packed fp32 fused multiply-adds on every CU (nothing else: no memory traffic); returns the achieved
 * TFLOP/s -- the arithmetic counterpart of lbm_copy_bandwidth, and what bench.py uses, after the copies, to bring the device to the
 * clocks of an arithmetic-bound load before the warm-up steps. */
int lbm_fma_rate(lbm_ctx* c, double ms, double* tflops);

#ifdef __cplusplus
}
#endif
#endif /* LBM_H */
