#!/bin/bash
# Reproduces the rocprofv3 summaries under profiles/ on an MI355X box (run from the repository root through gpurun):
#     gpurun --timeout 900 -- 'bash tools/profile_bench.sh <prefix> [bench.py arguments]'
# One --kernel-trace --stats pass and SEPARATE --pmc passes (never combined with a trace domain), the program itself after
# `--` (python3 bench.py ..., no wrapper), raw outputs under gpurun_out/prof/<prefix>/; the two small files that are committed,
#     profiles/<prefix>_kernel_stats.csv   profiles/<prefix>_pmc_summary.csv
# are written to gpurun_out/profiles/ (copy them into profiles/).
set -e
PREFIX=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof/$PREFIX
rm -rf "$OUT"; mkdir -p "$OUT" "$R/gpurun_out/profiles"
cd "$R"
ARGS="$*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py $ARGS --steps 640 --warmup 64 --no-cpu-baseline --no-extra > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err" || true
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o bench -- python3 bench.py $ARGS --steps 64 --warmup 8 --no-cpu-baseline --no-extra > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err" || true
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o bench -- python3 bench.py $ARGS --steps 64 --warmup 8 --no-cpu-baseline --no-extra > "$OUT/bench_sq.json" 2> "$OUT/bench_sq.err" || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_lds" -o bench -- python3 bench.py $ARGS --steps 64 --warmup 8 --no-cpu-baseline --no-extra > "$OUT/bench_lds.json" 2> "$OUT/bench_lds.err" || true
python3 tools/summarize_rocprof.py "$OUT" "$R/gpurun_out/profiles/$PREFIX" "tools/profile_bench.sh $PREFIX $ARGS"
tail -1 "$OUT/bench_trace.json" | cut -c1-160
head -4 "$R/gpurun_out/profiles/${PREFIX}_kernel_stats.csv"
