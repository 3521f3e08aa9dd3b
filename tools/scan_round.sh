#!/bin/bash
# int8 nomination: correctness (the search suite runs through it by default), phase stamps, then the headline timing
set -o pipefail
tag=${1:-r4x}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_search_gpu.py tests/test_search_fullsize_gpu.py tests/test_compact_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider -x > gpurun_out/${tag}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 25 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit 1
CODERAG_HIP_LIB=$GRAFT_REPO_ROOT/code-rag_amd/lib/libcoderag_hip_stamps.so timeout -k 10 300 python tools/fused_stamps.py > gpurun_out/${tag}_stamps.txt 2>&1 || exit 1
grep -A9 "^stamp" gpurun_out/${tag}_stamps.txt | grep -v "query image"; tail -n 8 gpurun_out/${tag}_stamps.txt; grep "end of the main loop per wave" -A2 gpurun_out/${tag}_stamps.txt | cut -c1-100
timeout -k 10 300 python bench.py --legs none --steps 50 --warmup 10 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -n 5 gpurun_out/${tag}_bench.err; exit 1; }
SCAN_TAG=$tag python - <<'PY'
import json
import os
d=json.load(open('gpurun_out/%s_bench.json' % os.environ.get('SCAN_TAG','r4x')))
print(d['search_stats']); print('value %.0f ms_per_step %.4f kernel_ms %.4f diff_us %.1f' % (d['value'], d['ms_per_step'],d['roofline']['kernel_ms'],(d['ms_per_step']-d['roofline']['kernel_ms'])*1e3), d['parity']['ids_bit_exact'], d['parity']['scores_bit_exact'])
PY
