#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile_gpu.sh r1_pmc'
# Pass 1: kernel trace + stats.  Passes 2..: one PMC group each (never mixed with the tracing domains).
# Output: gpurun_out/<tag>/{stats,fetch,write,mfma,lds}/...  + gpurun_out/<tag>/pmc_summary.json
set -eo pipefail
TAG=${1:-prof}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export PF_BENCH_GRAPH=0        # profile the eager launches (same kernels as the graph replay, one dispatch record each)
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $BENCH > "$OUT/bench_stats.log" 2>&1
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "busy:SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d "$OUT/$name" -o run -- $BENCH > "$OUT/bench_$name.log" 2>&1
    echo "pass $name done"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
# keep the merge-back small: the raw per-dispatch CSVs are summarised above
find "$OUT" -name "*.csv" -size +4M -delete
echo "profile $TAG complete"
