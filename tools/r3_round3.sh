#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider > gpurun_out/r3d_pytest.log 2>&1; echo "pytest rc=$?"
tail -n 12 gpurun_out/r3d_pytest.log
timeout -k 10 300 python bench.py --legs none --steps 50 --warmup 10 > gpurun_out/r3d_bench_headline.json 2> gpurun_out/r3d_bench_headline.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3d_bench_headline.json'))
print('ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'],'diff_us',(d['ms_per_step']-d['roofline']['kernel_ms'])*1e3, d['parity'])
PY
out=$PWD/gpurun_out/prof_r3d
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --legs none --steps 50 --warmup 10 --no-cpu-baseline --check-rows 0 > "$out/trace.log" 2>&1; echo "prof rc=$?"
find "$out" -name "*.csv" -size +20M -delete
python tools/summarize_prof.py $out gpurun_out/r3d_trace > /dev/null; head -n 20 gpurun_out/r3d_trace.md
timeout -k 10 900 python tools/store_scale_bench.py 10000000 > gpurun_out/r3d_store_scale_10M.json 2> gpurun_out/r3d_store_scale_10M.err; echo "store rc=$?"
cat gpurun_out/r3d_store_scale_10M.json
