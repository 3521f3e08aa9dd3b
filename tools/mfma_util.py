#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE pass:
utilisation = MFMA busy cycles / (kernel cycles x 256 CUs x 4 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is
summed over the 8 XCDs).  python tools/mfma_util.py <counter_collection.csv> <out.md>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
L = ["# MFMA utilisation (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE; bench.py through tools/gpu_prof.sh; the run's arguments are in profiles/README.md)", "",
     "utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs); means over the kernel's dispatches", "",
     "| kernel | dispatches | kernel cycles | MFMA busy cycles | MFMA utilisation |", "|---|---|---|---|---|"]
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or m.get("GRBM_GUI_ACTIVE", 0) <= 0:
        continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    name = k.split("(")[0].replace("void ", "")
    L.append(f"| `{name}` | {len(d['GRBM_GUI_ACTIVE'])} | {cyc:.0f} | {m['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} | {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:.1f} % |")
open(sys.argv[2], "w").write("\n".join(L) + "\n")
print("\n".join(L))
