#!/bin/bash
# Sequencer-side PMC passes (FIFO back-pressure, in-flight levels, instruction cache, waits) around a python script:
#   gpurun -- 'bash tools/pmc_sq.sh <tag> tools/tune_ec4.py'    -> gpurun_out/<tag>/pmc_summary.json
set -eo pipefail
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $SCRIPT "$@" > "$OUT/log_stats.txt" 2>&1
for pass in "fifo:SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
            "level:SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CYCLES" \
            "icache:SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_IFETCH SQ_LDS_ADDR_CONFLICT" \
            "wait:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
            "act:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
            "act2:SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU" \
            "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d "$OUT/$name" -o run -- python3 $SCRIPT "$@" > "$OUT/log_$name.txt" 2>&1 || echo "pass $name FAILED"
    echo "pass $name done"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
find "$OUT" -name "*.csv" -size +2M -delete
echo "pmc_sq $TAG complete"
