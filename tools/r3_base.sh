#!/bin/bash
# GPU box: round-3 baseline numbers (per-stage times at B = 4 / 8 / 32, kernel trace of the B = 4 step)
set -o pipefail
TAG=${1:-r3_base}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for B in 4 8 32; do timeout -k 10 200 python3 tools/stage_times.py $B > "$OUT/stage_b$B.txt" 2>&1; tail -20 "$OUT/stage_b$B.txt"; done
(cd /tmp && PF_BENCH_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_b4" -o run -- python3 $ROOT/bench.py --scaling strong --total-batch 4 --steps 50 --warmup 5 --no-cpu-baseline --no-reduced --no-pipelined > "$OUT/trace_b4.log" 2>&1)
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
echo "done $TAG"
