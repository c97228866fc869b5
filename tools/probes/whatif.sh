#!/bin/bash
# What bounds k_stream_walls at 4096^2 fp32?  Timing experiments with parts of the kernel compiled out (the results of these
# builds are GARBAGE: they answer "how long does the rest take", nothing else).  Builds variant libraries next to this script
# (tools/probes/_exp/, git-ignored) from lbm_streamw_f32.hip with the experiment macros of lbm_stream.hpp:
#   LBM_EXP_NO_COLLIDE          the update copies its inputs (memory + LDS + barriers remain)
#   LBM_EXP_NO_LDS              no posts, the update takes the row's own planes (memory + arithmetic + barriers remain)
#   LBM_EXP_NO_HBM              every segment works on the lattice's first 64 rows (cache resident)
#   LBM_EXPERIMENT_NO_BARRIER   the waves run free
# Usage (here, then on the GPU):  tools/probes/whatif.sh build ;  gpurun -- tools/probes/whatif.sh run
set -e
cd "$(dirname "$0")/../.."
CS=latticeboltzmannsimulations_amd/csrc
EXP=tools/probes/_exp
DT=${DT:-f32}            # DT=f64: the same for the double-precision kernel
VARIANTS="NO_COLLIDE:-DLBM_EXP_NO_COLLIDE NO_LDS:-DLBM_EXP_NO_LDS NO_HBM:-DLBM_EXP_NO_HBM NO_BARRIER:-DLBM_EXPERIMENT_NO_BARRIER NO_LDS_HBM:-DLBM_EXP_NO_LDS,-DLBM_EXP_NO_HBM VALU_ONLY:-DLBM_EXP_NO_LDS,-DLBM_EXP_NO_HBM,-DLBM_EXPERIMENT_NO_BARRIER"
if [ "$1" = build ]; then
    python -c "from latticeboltzmannsimulations_amd import _lib; _lib.build()"
    mkdir -p $EXP
    for v in $VARIANTS; do
        n=${v%%:*}; f=$(echo ${v#*:} | tr , ' ')
        (hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -w $f -c $CS/lbm_streamw_$DT.hip -o $EXP/sw_$n.o &&
         hipcc --offload-arch=gfx950 -fPIC -shared -o $EXP/lib_$n.so $EXP/sw_$n.o $(ls $CS/_obj/*.o | grep -v lbm_streamw_$DT) -ldl && rm $EXP/sw_$n.o) &
    done
    wait
    ls -la $EXP
else
    echo "== product"; python tools/perf_ab.py --steps 800 4096:$DT:fast 4096:$DT:strict 2>&1 | grep GLUPS
    for v in $VARIANTS; do
        n=${v%%:*}
        echo "== $n"; LBM_LIB_PATH=$PWD/$EXP/lib_$n.so timeout -k 10 100 python tools/perf_ab.py --steps 800 4096:$DT:fast 4096:$DT:strict 2>&1 | grep GLUPS
    done
fi
