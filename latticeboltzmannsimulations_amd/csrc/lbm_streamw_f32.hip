// lbm_streamw_f32.hip -- explicit instantiations of the streaming kernel with the walls inside, float (k_stream_walls, lbm_stream.hpp)
#define LBM_STREAMW_EXTERN
#define LBM_STREAM_ONLY_F32
#define LBM_STREAM_SKIP
#define LBM_STREAMP_SKIP
#define LBM_STREAMS_SKIP
#include "lbm_stream.hpp"
