#!/bin/bash
# headline + clustered + anisotropic legs at several sample sizes of the int8 scan's thresholds (CODERAG_HIP_I8_SAMPLE)
set -o pipefail
mkdir -p gpurun_out
for g in ${SWEEP:-8192 6144 4096 3072 8192}; do
  CODERAG_HIP_I8_SAMPLE=$g timeout -k 10 280 python bench.py --legs clustered,anisotropic --no-cpu-baseline --check-rows 0 > gpurun_out/i8s_$g.json 2> gpurun_out/i8s_$g.err || { tail -n 5 gpurun_out/i8s_$g.err; exit 1; }
  G=$g python - <<'PY'
import json, os
g = os.environ["G"]
d = json.load(open(f"gpurun_out/i8s_{g}.json"))
row = [f"sample {g:>5}: headline {d['ms_per_step']:.4f} ms (kernel {d['roofline']['kernel_ms']:.4f}, cands/q {d['search_stats']['candidates'] / d['search_stats']['batches'] / 64:.0f})"]
for k in ("clustered", "anisotropic"):
    r = d.get(k, {})
    if "ms_per_step" in r:
        row.append(f"{k} {r['ms_per_step']:.4f} (kernel {r['roofline']['kernel_ms']:.4f}, cands/q {r['candidates_per_query_and_batch']:.0f}, fb {r['fallback_used']})")
print("; ".join(row), flush=True)
PY
done
