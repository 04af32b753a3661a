#!/bin/bash
# GPU box: eval headline A/B over variant libraries: EVAL_VARIANTS="_a _b" bash tools/r5_evalab.sh [rounds] [extra bench args]
R=${1:-2}; shift
for r in $(seq $R); do
for v in "" $EVAL_VARIANTS; do
  lib=puflow_amd/libpuflow_hip$v.so
  [ -f $lib ] || continue
  ms=$(PF_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])") || exit 1
  echo "round $r lib$v ms_per_step patches/s $ms"
done
done
