#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [top]"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 18
for r in list(csv.DictReader(open(f)))[:top]:
    m = re.search(r"(k_\w+(<[\d, ]+>)?)", r["Name"])
    print(f"{(m.group(1) if m else r['Name'][:40]):28s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:9.1f} ms  {float(r['Percentage']):5.2f} %")
