#!/bin/bash
# GPU box, round 5: scratch job runner - each step under its own timeout, logs under gpurun_out/<tag>/
set -o pipefail
TAG=${1:-r5_job}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "=== step $i: $cmd" | tee -a "$OUT/steps.log"
  timeout -k 10 500 bash -c "$cmd" > "$OUT/step$i.log" 2>&1
  rc=$?
  grep -v amdgpu.ids "$OUT/step$i.log" | tail -${TAILN:-25}
  echo "=== step $i rc=$rc" | tee -a "$OUT/steps.log"
  if [ $rc -ge 124 ]; then echo "step $i timed out or was killed: stopping"; exit $rc; fi
done
