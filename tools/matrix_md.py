#!/usr/bin/env python3
"""matrix.json (tools/bench_matrix.py) -> markdown summary.  python tools/matrix_md.py in.json out.md"""
import json, sys
m = json.load(open(sys.argv[1]))
L = ["# Measurement matrix (SURVEY §8d), one MI355X, %d x %d — `tools/bench_matrix.py`, 10 warm-up + 50 timed calls each\n" % (m["rows"], m["dim"]),
     "| store | corpus | filter | queries/call | k | wall ms/call | device ms/call median (p10–p90) | scan kernel ms | scan GB/s |",
     "|---|---|---|---|---|---|---|---|---|"]
for r in m["search"]:
    d = r["call_ms_device"]
    L.append(f"| {r['store']} | {r['corpus']} | {r['filter'] or '—'} | {r['nq']} | {r['k']} | {r['wall_ms_per_call']:.3f} | "
             f"{d['median']:.3f} ({d['p10']:.3f}–{d['p90']:.3f}) | {r['scan_kernel_ms']:.3f} | {round(r['scan_GBps']) if r.get('scan_GBps') else '—'} |")
L += ["", "Calls with more than 64 queries are split into 64-query passes over the corpus (4 / 16 scans).", ""]
c = m.get("c5")
if c and "error" not in c:
    L += ["**Config 5 (hybrid re-rank), one GPU:** " + c["workload"] + f": scan + device re-rank **{c['scan_plus_rerank_ms_per_batch']:.3f} ms per batch** "
          f"({c['queries_per_s']:.0f} queries/s), of which gather + `crh_rerank_vector` + copy-back {c['rerank_ms_per_batch']:.3f} ms; the host `HybridRanker` "
          f"on the same candidate lists takes {c['host_hybrid_ranker_ms_per_batch']:.1f} ms per batch; survivors, order, f64 scores and signals identical: "
          f"{c['identical_to_host_ranker']}.", ""]
L += ["| encoder B | L | forward ms median (p10–p90) | chunks/s | TFLOP/s | of 2.5 PF |", "|---|---|---|---|---|---|"]
for r in m["encoder"]:
    d = r["forward_ms"]
    L.append(f"| {r['B']} | {r['L']} | {d['median']:.3f} ({d['p10']:.3f}–{d['p90']:.3f}) | {r['chunks_per_s']:.0f} | {r['TFLOPs']:.0f} | {r['frac_of_2.5PF'] * 100:.1f} % |")
open(sys.argv[2], "w").write("\n".join(L) + "\n")
print("\n".join(L))
