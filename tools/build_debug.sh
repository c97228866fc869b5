#!/bin/bash
# A debug build of the library (-DLBM_DEBUG: the two diagnostics LBM_DEBUG_SKIP_EXCHANGE / LBM_DEBUG_NO_EXCHANGE_READY become
# reachable through the environment) into scratch/liblbm_hip_debug.so; load it with LBM_LIB_PATH=<that file>.  Not shipped.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/latticeboltzmannsimulations_amd/csrc
O=$R/gpurun_debug; mkdir -p "$O"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DLBM_DEBUG"
for u in lbm_hip lbm_tiles_f32 lbm_tiles_f64 lbm_stream_f32 lbm_stream_f64; do
  /opt/rocm/bin/hipcc $F -c "$C/$u.hip" -o "$O/$u.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$O/liblbm_hip_debug.so" "$O"/*.o -ldl
ls -la "$O/liblbm_hip_debug.so"
