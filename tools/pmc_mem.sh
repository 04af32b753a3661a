#!/bin/bash
# Memory-path PMC passes (texture addresser, vector L1, L2) around a python script of this repo:
#   gpurun -- 'bash tools/pmc_mem.sh <tag> tools/tune_ec4.py'    -> gpurun_out/<tag>/pmc_summary.json
set -eo pipefail
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $SCRIPT "$@" > "$OUT/log_stats.txt" 2>&1
for pass in "ta:TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "tastall:TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "tcp2:TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
            "tcc:TCC_HIT_sum TCC_MISS_sum" "tcc2:TCC_REQ_sum TCC_BUSY_sum" "grbm:GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d "$OUT/$name" -o run -- python3 $SCRIPT "$@" > "$OUT/log_$name.txt" 2>&1 || echo "pass $name FAILED"
    echo "pass $name done"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
find "$OUT" -name "*.csv" -size +2M -delete
echo "pmc_mem $TAG complete"
