#!/bin/bash
# int8 nomination: correctness (the search suite runs through it by default), then the headline timing with and without it
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_search_gpu.py tests/test_search_fullsize_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider -x > gpurun_out/r3q_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 25 gpurun_out/r3q_pytest.log
[ $rc -eq 0 ] || exit 1
for i8 in 1 0; do
CODERAG_HIP_I8=$i8 timeout -k 10 300 python bench.py --legs none --steps 50 --warmup 10 > gpurun_out/r3q_bench_i8$i8.json 2> gpurun_out/r3q_bench_i8$i8.err || { tail -n 5 gpurun_out/r3q_bench_i8$i8.err; exit 1; }
python - $i8 <<'PY'
import json,sys
d=json.load(open('gpurun_out/r3q_bench_i8%s.json' % sys.argv[1]))
print('i8',sys.argv[1],'value %.0f ms_per_step %.4f kernel_ms %.4f diff_us %.1f' % (d['value'], d['ms_per_step'],d['roofline']['kernel_ms'],(d['ms_per_step']-d['roofline']['kernel_ms'])*1e3), d['parity']['ids_bit_exact'], d['parity']['scores_bit_exact'], d['search_stats'])
PY
done
