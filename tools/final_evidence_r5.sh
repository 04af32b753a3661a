#!/bin/bash
# GPU box: the round's final measurements (run through gpurun from the repo root, two calls - each stays under the 20-minute cap):
#   gpurun --timeout 1190 -- 'bash tools/final_evidence_r5.sh r5_final a'     bench lines of all four modes, batch sweep, unit timings
#   gpurun --timeout 1190 -- 'bash tools/final_evidence_r5.sh r5_final b'     kernel traces + PMC passes (headline, training step, CNF)
# Everything lands in gpurun_out/<tag>/; copy what is cited into profiles/<tag>/.
set -o pipefail
TAG=${1:-r5_final}
PART=${2:-a}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$PART" = a ]; then
  timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -1 "$OUT/bench.json" | cut -c1-200
  timeout -k 10 400 python3 bench.py --mode train --steps 100 --warmup 10 > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"; tail -1 "$OUT/bench_train.json" | cut -c1-200
  PF_TRAIN_PERSIST=0 timeout -k 10 300 python3 bench.py --mode train --steps 100 --warmup 10 --no-cpu-baseline --no-grad-parity > "$OUT/bench_train_per_layer.json" 2> /dev/null
  timeout -k 10 400 python3 bench.py --mode cnf --steps 10 --warmup 2 > "$OUT/bench_cnf.json" 2> "$OUT/bench_cnf.err"; tail -1 "$OUT/bench_cnf.json" | cut -c1-200
  timeout -k 10 400 python3 bench.py --mode pugan --steps 5 --warmup 2 > "$OUT/bench_pugan.json" 2> "$OUT/bench_pugan.err"; tail -1 "$OUT/bench_pugan.json" | cut -c1-200
  for B in 4 8 16 32; do
    timeout -k 10 120 python3 bench.py --scaling strong --total-batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-reduced --no-pipelined 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d' % $B, d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
  done > "$OUT/batch_sweep.txt"; cat "$OUT/batch_sweep.txt"
  timeout -k 10 200 python3 tools/time_ecunit.py 2>&1 | grep -v amdgpu.ids > "$OUT/time_ecunit.txt"
  timeout -k 10 200 python3 tools/time_emd.py 0 1 2 4 8 16 2>&1 | grep -v amdgpu.ids > "$OUT/emd_groups.txt"
  { timeout -k 10 120 python3 tools/time_fps.py 99840 20024 1,2,4,8,16,32,64 patch; timeout -k 10 120 python3 tools/time_fps.py 99840 20024 1,8,32 cube; } 2>&1 | grep -v amdgpu.ids > "$OUT/time_fps.txt"; tail -4 "$OUT/time_fps.txt"
else
  timeout -k 10 500 bash tools/profile_gpu.sh $TAG/pmc > "$OUT/profile_gpu.log" 2>&1; tail -1 "$OUT/profile_gpu.log"
  timeout -k 10 200 bash tools/train_prof.sh $TAG/train_trace > /dev/null 2>&1; head -3 "$OUT/train_trace/kernel_stats_per_step.txt"; tail -1 "$OUT/train_trace/kernel_stats_per_step.txt"
  timeout -k 10 400 bash tools/pmc_cmd.sh $TAG/train_pmc tools/train_eager_steps.py > "$OUT/train_pmc.log" 2>&1; tail -1 "$OUT/train_pmc.log"
  python3 tools/pmc_table.py "$OUT/train_pmc/pmc_summary.json" "rocprofv3 --pmc passes (tools/pmc_cmd.sh) over 4 EAGER training steps of BASELINE configs[2] (tools/train_eager_steps.py): per-kernel means" > "$OUT/train_pmc_table.txt"
  PF_TIME_CNF_ITERS=2 timeout -k 10 300 bash tools/pmc_cmd.sh $TAG/cnf_pmc tools/time_cnf.py > "$OUT/cnf_pmc.log" 2>&1; tail -1 "$OUT/cnf_pmc.log"
  python3 tools/pmc_table.py "$OUT/cnf_pmc/pmc_summary.json" "rocprofv3 --pmc passes over 2 forwards of bench.py --mode cnf's workload (tools/time_cnf.py): per-kernel means (no-op step attempts included in the step kernels' means)" > "$OUT/cnf_pmc_table.txt"
  timeout -k 10 300 bash tools/pmc_cmd.sh $TAG/fps_pmc tools/time_fps.py 99840 20024 32 patch > "$OUT/fps_pmc.log" 2>&1; tail -1 "$OUT/fps_pmc.log"
  find "$OUT" -name "*.csv" -size +2M -delete
fi
echo "evidence $TAG part $PART complete"
