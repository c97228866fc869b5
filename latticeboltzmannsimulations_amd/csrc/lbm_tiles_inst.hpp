// lbm_tiles_inst.hpp -- every instantiation of the multi-step tile kernel the dispatcher can ask for (lbm_host.hpp: dispatch();
// lbm_launch.hip: launch_deep()).  LBM_TILE_EXTERN is `extern` in the host units (declarations only: the kernels are
// compiled in lbm_tiles_f32.hip / lbm_tiles_f64.hip) and empty there (explicit instantiation definitions).
#ifndef LBM_TILE_EXTERN
#define LBM_TILE_EXTERN extern
#endif

#define LBM_TILE_ONE(R, COLL, SEM, S, TURB)                                                                                   \
    LBM_TILE_EXTERN template __global__ void k_stepS_deep<R, COLL, SEM, S, false, TURB>(                                      \
        const R* __restrict__, R* __restrict__, Geo, Relax<R>, Batch<R>, int, int, int, int, int, FramePtrs<R>, int, int, int, int, int);
// steps per launch 3 .. 5
#define LBM_TILE_S(R, COLL, SEM, TURB) LBM_TILE_ONE(R, COLL, SEM, 3, TURB) LBM_TILE_ONE(R, COLL, SEM, 4, TURB) LBM_TILE_ONE(R, COLL, SEM, 5, TURB)
// MRT_GPU semantics: six collision variants, with and without the closure; MRT.py semantics: the three strict ones
#define LBM_TILE_ALL(R)                                                                                                        \
    LBM_TILE_S(R, C_SRT, SEM_GPU, false) LBM_TILE_S(R, C_TRT, SEM_GPU, false) LBM_TILE_S(R, C_MRT, SEM_GPU, false)            \
    LBM_TILE_S(R, C_MRT_FAST, SEM_GPU, false) LBM_TILE_S(R, C_SRT_FAST, SEM_GPU, false) LBM_TILE_S(R, C_TRT_FAST, SEM_GPU, false) \
    LBM_TILE_S(R, C_SRT, SEM_GPU, true) LBM_TILE_S(R, C_TRT, SEM_GPU, true) LBM_TILE_S(R, C_MRT, SEM_GPU, true)               \
    LBM_TILE_S(R, C_MRT_FAST, SEM_GPU, true) LBM_TILE_S(R, C_SRT_FAST, SEM_GPU, true) LBM_TILE_S(R, C_TRT_FAST, SEM_GPU, true) \
    LBM_TILE_S(R, C_SRT, SEM_PY, false) LBM_TILE_S(R, C_TRT, SEM_PY, false) LBM_TILE_S(R, C_MRT, SEM_PY, false)

#if !defined(LBM_TILES_ONLY_F64)
LBM_TILE_ALL(float)
#endif
#if !defined(LBM_TILES_ONLY_F32)
LBM_TILE_ALL(double)
#endif
