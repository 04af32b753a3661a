#!/bin/bash
# GPU box: kernel trace of the captured training step (bench.py --mode train), per-step kernel table
set -o pipefail
TAG=${1:-r3_train}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o tr -- python3 $ROOT/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-grad-parity > "$OUT/trace.log" 2>&1)
python3 - "$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# 2 eager warm-up steps + 3 eager profiled steps (bench's call profile) + capture + 25 replays: count = calls of the one-per-step fused optimizer kernel
n = max(int(r["Calls"]) for r in rows if "adam_update_kernel" in r["Name"])
tot = 0.0
lines = []
for r in rows:
    per = float(r["TotalDurationNs"]) / n / 1e3
    tot += per
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:100]
    lines.append(f"{nm:100s} {int(r['Calls']) / n:6.1f}/step {per:8.1f} us/step")
open(out + "/kernel_stats_per_step.txt", "w").write(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-grad-parity   ({n} executions of the step's kernels)\n" + "\n".join(lines[:70]) + f"\ntotal {tot / 1e3:.2f} ms of kernel time per step, {sum(int(r['Calls']) for r in rows) / n:.0f} launches per step\n")
print("\n".join(lines[:45])); print(f"total {tot / 1e3:.2f} ms, launches {sum(int(r['Calls']) for r in rows) / n:.0f}")
PY
python3 - "$OUT" <<'PY'
# timeline of the LAST replayed step: wall, union of busy intervals, idle time, every launch in start order with its queue
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]), r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam_update_kernel" in r[2]]
# bench.py runs 3 EAGER steps (its call profile) after the timed replays: the last REPLAY ends at the fourth-last optimizer kernel
a, b = ends[-5] + 1, ends[-4] + 1
step = rows[a:b]
t0 = step[0][0]
busy, cur_s, cur_e = 0, step[0][0], step[0][1]
for s_, e_, _, _ in step[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
wall = max(r[1] for r in step) - t0
qs = sorted({r[3] for r in step})
with open(out + "/timeline_last_step.txt", "w") as fh:
    fh.write(f"last replayed step: {len(step)} launches, wall {wall / 1e3:.1f} us, some kernel running {busy / 1e3:.1f} us, nothing running {(wall - busy) / 1e3:.1f} us, queues {qs}\n")
    prev_end = t0
    for s_, e_, n_, q_ in step:
        fh.write(f"{(s_ - t0) / 1e3:9.1f} +{(e_ - s_) / 1e3:7.1f}  gap {(s_ - prev_end) / 1e3:6.1f}  q{qs.index(q_)}  {n_[:90]}\n")
        prev_end = max(prev_end, e_)
print(open(out + "/timeline_last_step.txt").readline())
PY
find "$OUT" -name "*kernel_trace.csv" -delete
echo "done $TAG"
