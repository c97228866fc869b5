// lbm_stream_f64.hip -- explicit instantiations of the strip-streaming multi-step kernel, double (see lbm_stream.hpp)
#define LBM_STREAM_EXTERN
#define LBM_STREAM_ONLY_F64
#define LBM_STREAMW_SKIP
#define LBM_STREAMP_SKIP
#define LBM_STREAMS_SKIP
#include "lbm_stream.hpp"
