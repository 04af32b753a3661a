#!/bin/bash
# GPU box: parity after the staging / tiling changes, stage times, kNN ablation at 4 patches, where the training step's torch launches come from
set -o pipefail
TAG=${1:-r3_c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; tail -4 "$OUT/pytest.log"
for B in 4 8 32; do timeout -k 10 200 python3 tools/stage_times.py $B 2>&1 | grep -v amdgpu.ids > "$OUT/stage_b$B.txt"; done
tail -16 "$OUT/stage_b4.txt"; tail -17 "$OUT/stage_b32.txt"
B=4 timeout -k 10 200 python3 tools/tune_knn.py 2>&1 | grep -v amdgpu.ids > "$OUT/knn_b4.txt"; cat "$OUT/knn_b4.txt"
timeout -k 10 300 python3 tools/op_sites.py 2>&1 | grep -v "amdgpu.ids\|Warn" > "$OUT/op_sites.txt"; head -30 "$OUT/op_sites.txt"
timeout -k 10 300 python3 tools/train_breakdown.py 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" > "$OUT/train_breakdown.txt"; cat "$OUT/train_breakdown.txt"
echo "done $TAG"
