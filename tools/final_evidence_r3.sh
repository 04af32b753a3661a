#!/bin/bash
# GPU box: the round's final measurements in one call (run through gpurun from the repo root):
#   gpurun --timeout 1190 -- 'bash tools/final_evidence_r3.sh r3_final'
# inference bench + kernel trace + PMC passes (tools/profile_gpu.sh), batch sweep, stage times, training step (bench line +
# kernel trace per step), continuous model bench line.  Everything lands in gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-r3_final}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -1 "$OUT/bench.json" | cut -c1-300
timeout -k 10 600 bash tools/profile_gpu.sh $TAG/pmc > "$OUT/profile_gpu.log" 2>&1; tail -2 "$OUT/profile_gpu.log"
for B in 4 8 16 32; do
  timeout -k 10 120 python3 bench.py --scaling strong --total-batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-reduced --no-pipelined 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d' % $B, d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done > "$OUT/batch_sweep.txt"; cat "$OUT/batch_sweep.txt"
for B in 4 32; do timeout -k 10 200 python3 tools/stage_times.py $B 2>&1 | grep -v amdgpu.ids > "$OUT/stage_b$B.txt"; done
timeout -k 10 300 python3 bench.py --mode train --steps 20 --warmup 5 > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"; tail -1 "$OUT/bench_train.json" | cut -c1-300
timeout -k 10 400 python3 bench.py --mode cnf --steps 5 --warmup 2 > "$OUT/bench_cnf.json" 2> "$OUT/bench_cnf.err"; tail -1 "$OUT/bench_cnf.json" | cut -c1-300
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -o tr -- python3 $ROOT/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/train_trace.log" 2>&1)
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cnf_trace" -o tr -- python3 $ROOT/bench.py --mode cnf --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/cnf_trace.log" 2>&1)
find "$OUT" -name "*kernel_trace.csv" -delete
timeout -k 10 300 python3 tools/replay_probe.py 4 2>&1 | grep -v amdgpu.ids > "$OUT/replay_probe_b4.txt"
echo "evidence $TAG complete"
