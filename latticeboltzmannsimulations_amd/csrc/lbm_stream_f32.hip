// lbm_stream_f32.hip -- explicit instantiations of the strip-streaming multi-step kernel, float (see lbm_stream.hpp)
#define LBM_STREAM_EXTERN
#define LBM_STREAM_ONLY_F32
#define LBM_STREAMW_SKIP
#define LBM_STREAMP_SKIP
#define LBM_STREAMS_SKIP
#include "lbm_stream.hpp"
