#!/bin/bash
# GPU box: does the host's OpenMP pool (CFS quota) show in the bench lines?  default threads vs OMP_NUM_THREADS=1
for r in 1 2 3; do
for t in "" 1; do
  ms=$(OMP_NUM_THREADS=$t timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])") || exit 1
  echo "round $r OMP_NUM_THREADS='$t' ms_per_step patches/s $ms"
done
done
