#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv: python tools/prof_feat_report.py <dir> <iterations> [top]"""
import csv, glob, sys
d, its = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:top]:
    print(f"{r['Name'][:84]:84s} {int(r['Calls']) / its:7.1f}/it avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['TotalDurationNs']) / its / 1e6:7.3f} ms/it")
print(f"total {tot / its / 1e6:.3f} ms/it, {sum(int(r['Calls']) for r in rows) / its:.0f} launches/it")
