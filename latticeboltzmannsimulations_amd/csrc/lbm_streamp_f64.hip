// lbm_streamp_f64.hip -- explicit instantiations of the streaming kernel with two rows per wave, double (k_stream_pairs, lbm_stream.hpp)
#define LBM_STREAMP_EXTERN
#define LBM_STREAM_ONLY_F64
#define LBM_STREAM_SKIP
#define LBM_STREAMW_SKIP
#define LBM_STREAMS_SKIP
#include "lbm_stream.hpp"
