#!/bin/bash
# PMC passes (one counter group per pass, never mixed with tracing domains other than --kernel-trace) around an
# arbitrary python script of this repo:   gpurun -- 'bash tools/pmc_cmd.sh <tag> tools/tune_ec4.py'
# Output: gpurun_out/<tag>/pmc_summary.json (per-kernel means of every counter; tools/pmc_summary.py)
set -eo pipefail
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $SCRIPT "$@" > "$OUT/log_stats.txt" 2>&1
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "issue:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
            "wait:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
            "coexec:SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA" \
            "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d "$OUT/$name" -o run -- python3 $SCRIPT "$@" > "$OUT/log_$name.txt" 2>&1
    echo "pass $name done"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
find "$OUT" -name "*.csv" -size +2M -delete
echo "pmc_cmd $TAG complete"
