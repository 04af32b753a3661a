#!/bin/bash
# Extra PMC passes for stall analysis of the dominant kernels (one counter group per pass).
#   gpurun -- 'bash tools/pmc_extra.sh <tag>'
set -eo pipefail
TAG=${1:-pmcx}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export PF_BENCH_GRAPH=0        # profile the eager launches (same kernels as the graph replay, one dispatch record each)
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
cd /tmp
for pass in "issue:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
            "wait:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
            "coexec:SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_BUSY_CYCLES" \
            "insts:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d "$OUT/$name" -o run -- $BENCH > "$OUT/bench_$name.log" 2>&1
    echo "pass $name done"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
find "$OUT" -name "*.csv" -size +4M -delete
echo "pmc_extra $TAG complete"
