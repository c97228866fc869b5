// lbm_tiles_f64.hip -- explicit instantiations of the multi-step tile kernel, double (see lbm_tiles_inst.hpp)
#include "lbm_kernels.hpp"
#define LBM_TILE_EXTERN
#define LBM_TILES_ONLY_F64
#include "lbm_tiles_inst.hpp"
