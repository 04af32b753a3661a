#!/bin/bash
# GPU box: kernel trace of the captured EVAL step at a small batch (default 4 patches = the per-GPU share of one 32-patch batch on
# 8 GPUs): every launch of one replayed step in start order with its duration and the idle gap in front of it.
#   gpurun -- 'bash tools/eval_timeline.sh r5_b4/timeline 4'
set -o pipefail
TAG=${1:-r5_b4/timeline}
BATCH=${2:-4}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o tr -- python3 $ROOT/bench.py --scaling strong --total-batch $BATCH --steps 50 --warmup 10 --no-cpu-baseline --no-reduced --no-pipelined > "$OUT/trace.log" 2>&1) || { tail -5 "$OUT/trace.log"; exit 1; }
python3 - "$OUT" "$BATCH" <<'PY'
import csv, glob, sys, re
out, batch = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])) for r in csv.DictReader(open(f))]
rows.sort()
# a step starts at its K = 16 neighbour search; the replayed steps are the back-to-back ones: take the window with the median period
starts = [i for i, r in enumerate(rows) if "knn5_kernel<16" in r[2]]
wins = [(rows[b][0] - rows[a][0], a, b) for a, b in zip(starts, starts[1:])]
wins.sort()
per, a, b = wins[len(wins) // 4]                       # lower quartile: a replay, not an eager step with host time in it
step = rows[a:b]
t0 = step[0][0]
lines, busy, gaps, prev_end = [], 0, 0, None
for s_, e_, nm in step:
    gap = 0 if prev_end is None else s_ - prev_end
    if prev_end is not None and gap > 0: gaps += gap
    busy += e_ - s_
    lines.append(f"{(s_ - t0) / 1e3:8.1f} + {(e_ - s_) / 1e3:6.1f}  gap {gap / 1e3:6.1f}  {nm[:110]}")
    prev_end = e_ if prev_end is None else max(prev_end, e_)
hdr = (f"rocprofv3 --kernel-trace -- python3 bench.py --scaling strong --total-batch {batch} --steps 50 --warmup 10 (one replayed step: start us + duration us, idle gap in front)\n"
       f"period {per / 1e3:.1f} us, {len(step)} launches, kernel time {busy / 1e3:.1f} us, idle between launches {gaps / 1e3:.1f} us (overlapping side-branch launches show negative gaps), "
       f"period minus last end {(per - (prev_end - t0)) / 1e3:.1f} us (replay-to-replay)\n")
open(out + "/timeline.txt", "w").write(hdr + "\n".join(lines) + "\n")
print(hdr + "\n".join(lines))
PY
rm -rf "$OUT/trace"
