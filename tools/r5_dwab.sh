#!/bin/bash
# GPU box: training step, persistent kernels x weight-gradient stream
for r in 1 2; do
for p in 1 0; do for w in 0 1; do
  out=$(PF_TRAIN_PERSIST=$p PF_TRAIN_DW_STREAM=$w timeout -k 10 200 python bench.py --mode train --steps 30 --warmup 8 --no-cpu-baseline --no-grad-parity 2>&1) || { echo "$out" | tail -5; exit 1; }
  ms=$(echo "$out" | grep '^{' | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  sk=$(echo "$out" | grep -c "skipped on the device")
  echo "round $r persist=$p dw_stream=$w ms_per_step $ms skipped_msgs $sk"
done; done; done
