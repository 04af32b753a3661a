#!/usr/bin/env python3
"""One line per kernel from a tools/pmc_summary.py JSON (rocprofv3 --pmc passes, tools/pmc_cmd.sh):
    python tools/pmc_table.py profiles/r3_final/train_pmc_summary.json [title] > train_pmc_table.txt
mfma  = matrix-pipe busy share: SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs).  `cycles` is GRBM_GUI_ACTIVE / 8 when that
        window is physical; when GRBM_GUI_ACTIVE / 8 / duration exceeds the part's 2.4 GHz (short kernels: the counter's window
        spans more than the kernel) it is duration x 2.4 GHz instead and the row carries a `*` - the uncorrected figure, which
        understates the utilisation by the same factor, is printed after it.
hbm   = FETCH_SIZE x 2 + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES;
ldsconf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; clk = GRBM_GUI_ACTIVE / 8 / duration (only where physical)."""
import json
import sys

NOMINAL_GHZ, SIMDS = 2.4, 1024


def main(path, title=""):
    d = json.load(open(path))
    ks = d.get("kernels", d)
    print(title or f"per-kernel means from {path} (tools/pmc_summary.py)")
    print(__doc__[__doc__.index("mfma  ="):].rstrip())
    for k, e in ks.items():
        if not e.get("avg_us") or not e.get("calls"):
            continue
        us = e["avg_us"]
        busy, gui = e.get("SQ_VALU_MFMA_BUSY_CYCLES_mean"), e.get("GRBM_GUI_ACTIVE_mean")
        mf = "      -"
        clk = "    -"
        if busy is not None and gui:
            ghz = gui / 8 / us / 1e3
            raw = 100.0 * busy / (gui / 8 * SIMDS)
            if ghz > NOMINAL_GHZ * 1.02:
                mf = f"{100.0 * busy / (us * 1e3 * NOMINAL_GHZ * SIMDS):5.1f}%* (gui window {raw:4.1f}%, {ghz:4.2f} GHz)"
            else:
                mf = f"{raw:5.1f}% "
                clk = f"{ghz:4.2f} GHz"
        hbm = e.get("hbm_bytes_per_launch")
        hb = f"{hbm / 1e6:7.1f} MB ({hbm / us / 1e6:5.2f} TB/s)" if hbm else "      -"
        wait = f"{100.0 * e['SQ_WAIT_ANY_mean'] / e['SQ_WAVE_CYCLES_mean']:3.0f}%" if e.get("SQ_WAVE_CYCLES_mean") else "  -"
        ldc = f"{e['lds_conflict_pct']:5.1f}%" if e.get("lds_conflict_pct") is not None else "    -"
        print(f"{k[:44]:44s} calls {e['calls']:4d} avg {us:7.1f} us  hbm {hb}  ldsconf {ldc}  wait {wait}  clk {clk}  mfma {mf}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
