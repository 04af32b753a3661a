#!/bin/bash
# GPU box: training step with / without the weight unit's first conv folded into its producers
for r in 1 2 3; do
for f in 1 0; do
  out=$(PF_TRAIN_FOLD_WU=$f timeout -k 10 200 python bench.py --mode train --steps 30 --warmup 8 --no-cpu-baseline --no-grad-parity 2>&1) || { echo "$out" | tail -5; exit 1; }
  echo "round $r fold=$f $(echo "$out" | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss'])")"
done; done
