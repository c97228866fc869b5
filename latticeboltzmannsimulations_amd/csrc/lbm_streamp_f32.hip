// lbm_streamp_f32.hip -- explicit instantiations of the streaming kernel with two rows per wave, float (k_stream_pairs, lbm_stream.hpp)
#define LBM_STREAMP_EXTERN
#define LBM_STREAM_ONLY_F32
#define LBM_STREAM_SKIP
#define LBM_STREAMW_SKIP
#define LBM_STREAMS_SKIP
#include "lbm_stream.hpp"
