// lbm_tiles_f32.hip -- explicit instantiations of the multi-step tile kernel, float (see lbm_tiles_inst.hpp)
#include "lbm_kernels.hpp"
#define LBM_TILE_EXTERN
#define LBM_TILES_ONLY_F32
#include "lbm_tiles_inst.hpp"
