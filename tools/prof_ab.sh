#!/bin/bash
# rocprofv3 counters of `tools/perf_ab.py <case>` for the library at $1 (A/B of two builds); output under gpurun_out/prof_ab/<tag>
set -e
LIB=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_ab/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export LBM_LIB_PATH=$LIB
cd "$R"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o p -- python3 tools/perf_ab.py --steps 160 --reps 1 "$@" > "$OUT/trace.log" 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o p -- python3 tools/perf_ab.py --steps 160 --reps 1 "$@" > "$OUT/sq.log" 2>&1 || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/pmc_lds" -o p -- python3 tools/perf_ab.py --steps 160 --reps 1 "$@" > "$OUT/lds.log" 2>&1 || true
python3 tools/summarize_rocprof.py "$OUT" "$OUT/summary"
cat "$OUT/summary_kernel_stats.csv" | head -5
grep k_stream "$OUT/summary_pmc_summary.csv"
