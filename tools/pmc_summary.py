#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/profile_gpu.sh: per-kernel mean of every collected counter
(+ derived HBM bytes with the gfx950 FETCH_SIZE correction, MFMA utilisation, LDS conflict rate)
and the kernel-trace average durations.  Usage: pmc_summary.py gpurun_out/<tag> > pmc_summary.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

SIMD_PER_CU, CUS, XCDS = 4, 256, 8
NOMINAL_GHZ = 2.4            # MI355X_MICROARCH.md: peak engine clock


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name).strip()


def main(root: str) -> None:
    ctr = defaultdict(lambda: defaultdict(list))        # kernel -> counter -> values
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                dur[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                             pct=float(r["Percentage"]))
    out = {}
    for k in sorted(set(ctr) | set(dur), key=lambda k: -dur.get(k, {}).get("pct", 0.0)):
        e = dict(dur.get(k, {}))
        for c, v in ctr.get(k, {}).items():
            e[c + "_mean"] = sum(v) / len(v)
            e["launches_" + c] = len(v)
        if "FETCH_SIZE_mean" in e and "WRITE_SIZE_mean" in e:
            # units KB; FETCH_SIZE under-counts 16 B/lane loads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section)
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_mean"] + e["WRITE_SIZE_mean"]) * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES_mean" in e and e.get("GRBM_GUI_ACTIVE_mean"):
            # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs (value / 8 / duration = the ~2.1-2.5 GHz shader clock);
            # MfmaUtil = busy cycles / (active cycles x SIMDs), as in counter_defs.yaml with reduce(max) over XCDs
            gui = e["GRBM_GUI_ACTIVE_mean"] / XCDS
            e["mfma_util_pct"] = 100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / (gui * SIMD_PER_CU * CUS)
            if e.get("avg_us"):
                e["shader_clock_ghz"] = gui / e["avg_us"] / 1e3
                # GRBM_GUI_ACTIVE counts from before the dispatch's first wave to after its last: for a short kernel the window is
                # longer than the kernel and "GUI cycles / duration" comes out above the 2.4 GHz the part can clock.  Such a window
                # UNDERSTATES every per-cycle ratio; the utilisation is then taken against duration x the nominal clock (an upper
                # bound on the cycles the kernel really had) and the row is flagged.
                if e["shader_clock_ghz"] > NOMINAL_GHZ * 1.02:
                    e["mfma_util_pct_gui_window"] = e["mfma_util_pct"]
                    e["mfma_util_pct"] = 100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / (e["avg_us"] * 1e3 * NOMINAL_GHZ * SIMD_PER_CU * CUS)
                    e["clock_unphysical"] = True
                    e["shader_clock_ghz_gui_window"] = e["shader_clock_ghz"]
                    e["shader_clock_ghz"] = None
        if e.get("SQ_LDS_IDX_ACTIVE_mean"):
            e["lds_conflict_pct"] = 100.0 * e.get("SQ_LDS_BANK_CONFLICT_mean", 0.0) / e["SQ_LDS_IDX_ACTIVE_mean"]
        out[k] = e
    json.dump(dict(how="tools/profile_gpu.sh: rocprofv3 --kernel-trace --stats, then one --pmc group per pass "
                       "(python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline, 32 x 2048-pt patches); counters are "
                       "per-dispatch sums over all XCDs as rocprofv3 reports them; FETCH_SIZE doubled per the gfx950 note",
                   kernels=out), sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
