#!/bin/bash
# GPU box: training step with an environment variable at two values:  bash tools/r5_envab2.sh VAR a b [rounds]
V=$1; A=$2; Bv=$3; R=${4:-3}
for r in $(seq $R); do
for f in $A $Bv; do
  out=$(env $V=$f timeout -k 10 200 python bench.py --mode train --steps 30 --warmup 8 --no-cpu-baseline --no-grad-parity 2>&1) || { echo "$out" | tail -5; exit 1; }
  echo "round $r $V=$f $(echo "$out" | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss'])")"
done; done
