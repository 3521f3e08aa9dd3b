#!/bin/bash
# round-3 first GPU pass: tests (no -x), skinny-vs-mid latency at T = 16, default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r3a_pytest.log 2>&1; echo "pytest rc=$?"
tail -n 40 gpurun_out/r3a_pytest.log
timeout -k 10 200 python tools/latency_bench.py > gpurun_out/r3a_latency_skinny.log 2>&1; echo "lat rc=$?"
CODERAG_HIP_SKINNY=0 timeout -k 10 200 python tools/latency_bench.py > gpurun_out/r3a_latency_noskinny.log 2>&1; echo "lat2 rc=$?"
paste gpurun_out/r3a_latency_skinny.log gpurun_out/r3a_latency_noskinny.log
timeout -k 10 600 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"
tail -n 30 gpurun_out/r3a_bench.err
