#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of any python script of this repo -> gpurun_out/<tag>/kernel_stats.txt (top kernels)
#   gpurun -- 'bash tools/prof_script.sh <tag> tools/time_ecunit.py [args]'
set -o pipefail
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o tr -- python3 $SCRIPT "$@" > "$OUT/run.log" 2>&1)
python3 - "$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(out + "/kernel_stats.txt", "w") as fh:
    for r in rows[:40]:
        nm = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:90]
        fh.write(f"{nm:90s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}  max {float(r['MaxNs']) / 1e3:8.1f}  {float(r['Percentage']):5.1f} %\n")
print(open(out + "/kernel_stats.txt").read())
PY
find "$OUT" -name "*kernel_trace.csv" -delete
