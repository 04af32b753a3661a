#!/bin/bash
# GPU box: rocprofv3 kernel trace of a python script -> gpurun_out/<tag>/sequence.txt, the LAST <n> kernel launches in order,
# run-length compressed (name x count, total us, and the idle gap in front of the run)
#   gpurun -- 'bash tools/trace_sequence.sh <tag> <n> tools/time_cnf.py'
set -o pipefail
TAG=$1; N=$2; shift; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o tr -- python3 $SCRIPT "$@" > "$OUT/run.log" 2>&1)
python3 - "$OUT" "$N" <<'PY'
import csv, glob, re, sys
out, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
runs = []
prev_end = None
for r in rows:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:80]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    if runs and runs[-1][0] == nm:
        runs[-1][1] += 1; runs[-1][2] += e - s; runs[-1][3] += gap
    else:
        runs.append([nm, 1, e - s, gap])
    prev_end = e
with open(out + "/sequence.txt", "w") as fh:
    for nm, c, d, g in runs:
        fh.write(f"{nm:80s} x{c:4d}  busy {d / 1e3:9.1f} us  gaps {g / 1e3:8.1f} us\n")
    fh.write(f"span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3:.1f} us over {len(rows)} launches\n")
PY
find "$OUT" -name "*kernel_trace.csv" -delete
