#!/bin/bash
# GPU box: the round's final measurements in one call (run through gpurun from the repo root):
#   gpurun --timeout 1190 -- 'bash tools/final_evidence.sh r2_final2'
# inference bench + kernel trace + PMC passes (tools/profile_gpu.sh), batch sweep, training step (eager / graphed, kernel
# trace, section breakdown), continuous model.  Everything lands in gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-final}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" ; tail -1 "$OUT/bench.json" | cut -c1-400
timeout -k 10 300 python3 bench.py --mode train --steps 20 --warmup 5 > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"; tail -1 "$OUT/bench_train.json" | cut -c1-300
timeout -k 10 600 bash tools/profile_gpu.sh $TAG/pmc > "$OUT/profile_gpu.log" 2>&1; tail -2 "$OUT/profile_gpu.log"
for B in 4 8 16 32; do
  timeout -k 10 120 python3 bench.py --scaling strong --total-batch $B --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d' % $B, d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done > "$OUT/batch_sweep.txt"
for B in 4 8 16 32; do
  timeout -k 10 120 python3 bench.py --scaling strong --total-batch $B --pipeline 3 --steps 60 --warmup 6 --no-cpu-baseline --no-reduced 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d, 3 steps in flight' % $B, d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done >> "$OUT/batch_sweep.txt"; cat "$OUT/batch_sweep.txt"
timeout -k 10 200 python3 tools/time_train.py > "$OUT/time_train.txt" 2>&1; cat "$OUT/time_train.txt"
timeout -k 10 200 python3 tools/train_breakdown.py 2>&1 | grep -v "Warn\|warn" > "$OUT/train_breakdown.txt"; cat "$OUT/train_breakdown.txt"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -o tr -- python3 $ROOT/tools/prof_train.py > "$OUT/train_trace.log" 2>&1)
find "$OUT/train_trace" -name "*kernel_trace.csv" -delete
timeout -k 10 200 python3 tools/time_cnf.py > "$OUT/time_cnf.txt" 2>&1; head -4 "$OUT/time_cnf.txt"
timeout -k 10 200 python3 tools/stage_times.py > "$OUT/stage_times.txt" 2>&1; tail -15 "$OUT/stage_times.txt"
timeout -k 10 200 python3 tools/time_fps.py 2>&1 | grep "B=" > "$OUT/time_fps.txt"; for b in 16 32; do timeout -k 10 100 python3 tools/time_fps.py 99840 20024 $b 2>&1 | grep "B=" >> "$OUT/time_fps.txt"; done; cat "$OUT/time_fps.txt"
timeout -k 10 200 python3 tools/time_patch.py 2>&1 | grep -v amdgpu.ids > "$OUT/time_patch.txt"; cat "$OUT/time_patch.txt"
timeout -k 10 300 python3 tools/time_cli.py 5000 64 2>&1 | grep -v amdgpu.ids > "$OUT/time_cli.txt"; cat "$OUT/time_cli.txt"
echo "evidence $TAG complete"
