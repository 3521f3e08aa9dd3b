#!/usr/bin/env python3
"""Timeline of the int8-nominated batches from a rocprofv3 --kernel-trace CSV: for the steady-state batches, when each kernel of a
batch starts and ends relative to the end of the previous batch's pass (k_scan_i8<..., 3>), per queue.
python tools/i8_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, os, re
import numpy as np
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
def short(n):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]
def is_pass(n):
    m = re.search(r"k_scan_i8<([^>]*)>", n)
    if not m:
        return False
    a = [x.strip() for x in m.group(1).split(",")]
    return len(a) == 4 or a[4] in ("0", "3")
passes = [i for i, r in enumerate(rows) if is_pass(r[2])]
print(f"{len(rows)} dispatches, {len(passes)} passes")
# steady state: the middle passes
use = passes[len(passes) // 3: 2 * len(passes) // 3]
acc = {}
for pi in use:
    t_end_prev = rows[pi][0]            # start of this pass
    prev = [p for p in passes if p < pi]
    if not prev:
        continue
    pe = rows[prev[-1]][1]              # end of the previous pass
    for r in rows[prev[-1] + 1: pi + 1]:
        acc.setdefault((short(r[2]), r[3]), []).append(((r[0] - pe) / 1e3, (r[1] - pe) / 1e3))
print("kernel (queue): start / end in us after the END of the previous pass, median over the steady-state batches")
for (name, q), v in sorted(acc.items(), key=lambda kv: np.median([a for a, _ in kv[1]])):
    a = np.array(v)
    print(f"  {name:40s} q{q:>3}  n={len(v):3d}  start {np.median(a[:,0]):8.1f}  end {np.median(a[:,1]):8.1f}  dur {np.median(a[:,1]-a[:,0]):7.1f}")
d = np.array([rows[p][1] - rows[p][0] for p in use]) / 1e3
g = np.array([rows[b][0] - rows[a][1] for a, b in zip(use[:-1], use[1:])]) / 1e3
print(f"pass duration median {np.median(d):.1f} us; gap between passes median {np.median(g):.1f} us; period {np.median(d) + np.median(g):.1f} us")
