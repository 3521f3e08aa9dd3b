#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_search_gpu.py tests/test_search_fullsize_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r3i_pytest.log 2>&1; echo "pytest rc=$?"
tail -n 8 gpurun_out/r3i_pytest.log
for fused in 1 0 1 0; do
CODERAG_HIP_FUSED_SCAN=$fused timeout -k 10 300 python bench.py --legs none --steps 50 --warmup 10 > gpurun_out/r3i_bench_f$fused.json 2> gpurun_out/r3i_bench_f$fused.err; echo "bench fused=$fused rc=$?"
python - $fused <<'PY'
import json,sys
d=json.load(open('gpurun_out/r3i_bench_f%s.json' % sys.argv[1]))
print('fused',sys.argv[1],'ms_per_step %.4f kernel_ms %.4f diff_us %.1f frac %.4f' % (d['ms_per_step'],d['roofline']['kernel_ms'],(d['ms_per_step']-d['roofline']['kernel_ms'])*1e3, d['roofline']['frac']), d['parity']['ids_bit_exact'], d['parity']['scores_bit_exact'])
PY
done
