#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_search_gpu.py tests/test_search_fullsize_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r3p_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 5 gpurun_out/r3p_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --legs none --steps 50 --warmup 10 > gpurun_out/r3p_bench.json 2> gpurun_out/r3p_bench.err || exit 1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3p_bench.json'))
print('ms_per_step %.4f kernel_ms %.4f diff_us %.1f frac %.4f' % (d['ms_per_step'],d['roofline']['kernel_ms'],(d['ms_per_step']-d['roofline']['kernel_ms'])*1e3, d['roofline']['frac']), d['parity']['ids_bit_exact'], d['parity']['scores_bit_exact'])
PY
