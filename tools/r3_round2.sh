#!/bin/bash
# round-3 second GPU pass: tests (no -x), store host side at 10M rows (upsert, compaction, snapshot), default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider > gpurun_out/r3b_pytest.log 2>&1; echo "pytest rc=$?"
tail -n 30 gpurun_out/r3b_pytest.log
timeout -k 10 900 python tools/store_scale_bench.py 10000000 > gpurun_out/r3b_store_scale_10M.json 2> gpurun_out/r3b_store_scale_10M.err; echo "store rc=$?"
tail -n 5 gpurun_out/r3b_store_scale_10M.err; cat gpurun_out/r3b_store_scale_10M.json
timeout -k 10 600 python bench.py > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; echo "bench rc=$?"
tail -n 30 gpurun_out/r3b_bench.err
