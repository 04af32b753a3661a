#!/usr/bin/env python3
"""Per-SIMD issue budget of the eval kernels from a tools/pmc_sq.sh summary:
    python tools/pmc_budget.py gpurun_out/<tag>/pmc_sq/pmc_summary.json > profiles/<round>/issue_budget.txt
Every share is a fraction of the kernel's cycles on ONE SIMD (GRBM_GUI_ACTIVE / 8 XCDs = the kernel's cycles; SQ counters are
sums over the chip's 1024 SIMDs; the SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES family counts in units of 4 cycles):
  mfma   = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / cycles          matrix pipe busy
  valu   = 4 SQ_ACTIVE_INST_VALU / 1024 / cycles             a vector instruction (MFMA issue included) being issued
  lds / vmem / salu = the same for SQ_ACTIVE_INST_LDS / _VMEM / _SCA
  any    = 4 SQ_ACTIVE_INST_ANY / 1024 / cycles              some instruction of some wave being issued
  coexec = SQ_VALU_MFMA_COEXEC_CYCLES / 1024 / cycles        vector instructions executing WHILE the matrix pipe is busy
  waves  = 4 SQ_WAVE_CYCLES / 1024 / cycles                  resident waves per SIMD, time average
  wait   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                 share of a wave's life spent waiting for an instruction to be issuable
  waitlds= SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  tafull = 4 SQ_VMEM_TA_CMD_FIFO_FULL / 1024 / cycles        vector-memory issue blocked by the texture addresser's queue
  VALU per MFMA-free slot: (SQ_INSTS_VALU - MFMAs) is not available per kernel here; see tools/isa_count.py"""
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
NS = 1024.0
hdr = f"{'kernel':44s} {'us':>7s} {'GHz':>5s} {'mfma':>6s} {'valu':>6s} {'lds':>6s} {'vmem':>6s} {'salu':>6s} {'any':>6s} {'coexec':>6s} {'waves':>6s} {'wait':>6s} {'waitlds':>7s} {'tafull':>6s}"
print(hdr)
for k, e in d.items():
    if "GRBM_GUI_ACTIVE_mean" not in e or e.get("pct", 0) < 1.0:
        continue
    us = e["avg_us"]
    cyc = e["GRBM_GUI_ACTIVE_mean"] / 8.0
    if cyc / us / 1e3 > 2.45:                      # the counter's window is longer than a short kernel: duration x 2.4 GHz instead
        cyc = us * 1e3 * 2.4
        ghz = "  -  "
    else:
        ghz = f"{cyc / us / 1e3:5.2f}"
    g = lambda n: e.get(n + "_mean", 0.0)
    sh = lambda v: 100.0 * v / NS / cyc
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    print(f"{k[:44]:44s} {us:7.1f} {ghz} {sh(g('SQ_VALU_MFMA_BUSY_CYCLES')):5.1f}% {sh(4 * g('SQ_ACTIVE_INST_VALU')):5.1f}% "
          f"{sh(4 * g('SQ_ACTIVE_INST_LDS')):5.1f}% {sh(4 * g('SQ_ACTIVE_INST_VMEM')):5.1f}% {sh(4 * g('SQ_ACTIVE_INST_SCA')):5.1f}% "
          f"{sh(4 * g('SQ_ACTIVE_INST_ANY')):5.1f}% {sh(g('SQ_VALU_MFMA_COEXEC_CYCLES')):5.1f}% {4 * wc / NS / cyc:6.2f} "
          f"{100 * g('SQ_WAIT_INST_ANY') / wc:5.1f}% {100 * g('SQ_WAIT_INST_LDS') / wc:6.1f}% {sh(4 * g('SQ_VMEM_TA_CMD_FIFO_FULL')):5.1f}%")
