#!/usr/bin/env python3
"""Instruction-class census of one kernel's gfx950 ISA (hipcc -S): MFMA / VALU / SALU / LDS / VMEM counts and the top VALU
opcodes.  python tools/isa_count.py <source.hip> <mangled-name-substring> [...]   (build container, no GPU)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from puflow_amd import build as B

src = sys.argv[1]
path = src if os.path.exists(src) else os.path.join(B.CSRC, src)
asm = os.path.join(tempfile.gettempdir(), os.path.basename(path) + ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(os.path.basename(path), []) +
                      ["-S", "--cuda-device-only", "-o", asm, path], stderr=subprocess.DEVNULL)
text = open(asm).read()
for pat in sys.argv[2:]:
    for m in re.finditer(r"^(_Z\S*" + re.escape(pat) + r"\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        cnt, valu = collections.Counter(), collections.Counter()
        for line in m.group(2).splitlines():
            line = line.strip()
            if not line or line[0] in ";." or line.endswith(":"):
                continue
            op = line.split()[0]
            k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_")
                 else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "other")
            cnt[k] += 1
            if k == "valu":
                valu[op] += 1
        print(subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip())
        print("  ", dict(cnt))
        print("  ", ", ".join(f"{k} {v}" for k, v in valu.most_common(16)))
