#!/bin/bash
# GPU box: rounds / time of the merge on the pipeline's own clouds for FPS variant libraries
for v in "" $FPS_VARIANTS; do
  lib=puflow_amd/libpuflow_hip$v.so
  [ -f $lib ] || continue
  echo "--- $lib"
  PF_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/fps_rounds_probe.py 4 2>&1 | grep "^cloud" || exit 1
  PF_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --mode pugan --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pugan', round(d['clouds_per_s'],1), 'clouds/s', d['stage_ms']['fps_merge'], 'ms merge')" || exit 1
done
