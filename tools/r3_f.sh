#!/bin/bash
# GPU box: the whole GPU test suite, then the three bench modes
set -o pipefail
TAG=${1:-r3_f}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; tail -6 "$OUT/pytest.log"
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -1 "$OUT/bench.json" | cut -c1-600
timeout -k 10 400 python3 bench.py --mode train --steps 20 --warmup 5 > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"; tail -1 "$OUT/bench_train.json" | cut -c1-1500
timeout -k 10 400 python3 bench.py --mode cnf --steps 5 --warmup 2 > "$OUT/bench_cnf.json" 2> "$OUT/bench_cnf.err"; tail -1 "$OUT/bench_cnf.json" | cut -c1-1800
for B in 4 8 16 32; do
  timeout -k 10 120 python3 bench.py --scaling strong --total-batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-reduced --no-pipelined 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d' % $B, d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done > "$OUT/batch_sweep.txt"; cat "$OUT/batch_sweep.txt"
echo "done $TAG"
