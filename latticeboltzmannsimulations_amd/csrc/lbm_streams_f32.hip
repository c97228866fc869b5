// lbm_streams_f32.hip -- explicit instantiations of the streaming kernel with the walls inside for a slab, float (k_stream_walls_slab, lbm_stream.hpp)
#define LBM_STREAMS_EXTERN
#define LBM_STREAM_ONLY_F32
#define LBM_STREAM_SKIP
#define LBM_STREAMW_SKIP
#define LBM_STREAMP_SKIP
#include "lbm_stream.hpp"
