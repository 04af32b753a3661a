#!/usr/bin/env python3
"""Build container: timing-only variants of the persistent EdgeConv forward kernel (-DPF_ECP_DBG=mask, csrc/train_fused.hip);
GPU box: time the 128-channel unit's forward with each (the kernel's own duration from rocprofv3 would be better still; the
HIP-event time of the whole call is enough to rank the pieces).
  python tools/ecp_ablate.py build            (here)
  python tools/ecp_ablate.py run              (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MASKS = (0, 1, 2, 4, 8, 16, 1 | 2, 1 | 2 | 4 | 8 | 16)
if sys.argv[1] == "build":
    from puflow_amd import build as B
    for m in MASKS[1:]:
        print(B.build(verbose=False, defines=[f"PF_ECP_DBG={m}"], tag=f"ecp{m}", only=("train_fused.hip",)))
else:
    from puflow_amd.build import LIB
    for m in MASKS:
        lib = LIB if m == 0 else LIB.replace(".so", f"_ecp{m}.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_ecunit.py")], env=dict(os.environ, PF_LIB_PATH=lib, PF_ECUNIT_ONLY="1"),
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if "persistent=1" in l]
        print(f"mask {m:2d}: {line[0] if line else r.stderr[-300:]}", flush=True)
