#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch report of every HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Fails if any kernel uses scratch: a spilled / stack-bounced value in these MFMA kernels is always a performance bug
(and one such pattern misbehaved on hardware during round 1).   python tools/check_resources.py [--write profiles/...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from puflow_amd import build as B

rows = []
for src in B.SOURCES:
    path = os.path.join(B.CSRC, src)
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(os.path.basename(path), []) + ["-c", path, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"file": src, "kernel": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        else:
            cur[k.split(" ")[0]] = int(v)
lines = ["| file | kernel | VGPR | AGPR | scratch B/lane | waves/SIMD | LDS B |", "|---|---|---|---|---|---|---|"]
bad = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["kernel"]).split("(")[0]
    lines.append(f"| {r['file']} | `{name}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('ScratchSize')} | {r.get('Occupancy')} | {r.get('LDS')} |")
    if r.get("ScratchSize", 0) > 0:
        bad.append(name)
text = "\n".join(lines)
print(text)
if "--write" in sys.argv:
    open(sys.argv[sys.argv.index("--write") + 1], "w").write("# hipcc kernel resource usage (gfx950)\n\n" + text + "\n")
if bad:
    raise SystemExit(f"kernels using scratch: {bad}")
