#!/bin/bash
# GPU box: the user-facing flow end to end - train 11 epochs on synthetic patches (graph-replayed steps, validation CD must
# move), the checkpoint the run writes, the .xyz CLI on three clouds with that checkpoint.    bash tools/e2e_smoke.sh
set -eo pipefail
D=${1:-/tmp/e2e}
rm -rf "$D" && mkdir -p "$D/in" "$D/out"
timeout -k 10 400 python3 -m puflow_amd.train --synthetic --max_epochs 11 --val_batches 2 --graph --checkpoint_path "$D/model.ckpt" > "$D/train.out" 2> "$D/train.err" || { tail -5 "$D/train.err"; exit 1; }
grep "epoch 0:\|epoch 10:\|saved" "$D/train.out"
python3 - "$D" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from puflow_amd.weights import synth_patches
for k in range(3):
    np.savetxt(f"{sys.argv[1]}/in/c{k}.xyz", synth_patches(1, 2048, seed=k)[0].numpy() * 2 + 1, fmt="%.6f")
PY
timeout -k 10 300 python3 -m puflow_amd.upsample --source "$D/in" --target "$D/out" --checkpoint "$D/model-epoch11.ckpt" --up_ratio 4 2> "$D/cli.err" || { tail -5 "$D/cli.err"; exit 1; }
python3 - "$D" <<'PY'
import sys, glob, numpy as np
files = sorted(glob.glob(f"{sys.argv[1]}/out/*.xyz"))
assert len(files) == 3, files
for f in files:
    a = np.loadtxt(f)
    assert a.shape == (8192, 3) and np.isfinite(a).all(), (f, a.shape)
    print(f, a.shape, "finite", a.min(0).round(2), a.max(0).round(2))
print("e2e smoke ok")
PY
