#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_search_gpu.py tests/test_search_fullsize_gpu.py tests/test_store_gpu.py tests/test_compact_gpu.py tests/test_snapshot_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r3e_pytest.log 2>&1; echo "pytest rc=$?"
tail -n 8 gpurun_out/r3e_pytest.log
timeout -k 10 300 python bench.py --legs none --steps 50 --warmup 10 > gpurun_out/r3e_bench_headline.json 2> gpurun_out/r3e_bench_headline.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3e_bench_headline.json'))
print('ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'],'diff_us',(d['ms_per_step']-d['roofline']['kernel_ms'])*1e3, d['parity'])
PY
out=$PWD/gpurun_out/prof_r3e
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --legs none --steps 50 --warmup 10 --no-cpu-baseline --check-rows 0 > "$out/trace.log" 2>&1; echo "prof rc=$?"
find "$out" -name "*.csv" -size +20M -delete
python tools/summarize_prof.py $out gpurun_out/r3e_trace > /dev/null; head -n 16 gpurun_out/r3e_trace.md
