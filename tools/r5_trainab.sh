#!/bin/bash
# GPU box: training-step A/B over variant libraries: TRAIN_VARIANTS="_a _b" bash tools/r5_trainab.sh [rounds]
R=${1:-2}
for r in $(seq $R); do
for v in "" $TRAIN_VARIANTS; do
  lib=puflow_amd/libpuflow_hip$v.so
  [ -f $lib ] || continue
  ms=$(PF_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --mode train --steps 30 --warmup 8 --no-cpu-baseline --no-grad-parity 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])") || exit 1
  echo "round $r lib$v ms_per_step $ms"
done
done
