#!/bin/bash
# GPU box, round 5, first call: this round's baseline on one box + the issue-side PMC budget of the eval kernels
#   gpurun --timeout 1190 -- 'bash tools/r5_base.sh r5_base'
set -o pipefail
TAG=${1:-r5_base}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --no-reduced > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -1 "$OUT/bench.json" | cut -c1-160
timeout -k 10 300 python3 bench.py --mode train --steps 100 --warmup 10 --no-cpu-baseline --no-grad-parity > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"; tail -1 "$OUT/bench_train.json" | cut -c1-160
for B in 4 8; do
  timeout -k 10 120 python3 bench.py --scaling strong --total-batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-reduced --no-pipelined 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d' % $B, d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done > "$OUT/batch_sweep.txt"; cat "$OUT/batch_sweep.txt"
timeout -k 10 120 python3 tools/stage_times.py 4 2048 2>&1 | grep -v amdgpu.ids > "$OUT/stage_b4.txt"
timeout -k 10 120 python3 tools/stage_times.py 32 2048 2>&1 | grep -v amdgpu.ids > "$OUT/stage_b32.txt"
timeout -k 10 200 python3 tools/tune_ec4.py abl 2>&1 | grep -v amdgpu.ids > "$OUT/tune_ablation.txt"; tail -12 "$OUT/tune_ablation.txt"
echo "base done"
PF_BENCH_GRAPH=0 timeout -k 10 600 bash tools/pmc_sq.sh $TAG/pmc_sq bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reduced --no-pipelined > "$OUT/pmc_sq.log" 2>&1; tail -2 "$OUT/pmc_sq.log"
echo "r5_base complete"
