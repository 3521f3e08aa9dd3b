#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
for lib in code-rag_amd/lib/variants/*.so; do
  CODERAG_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --legs none --steps 50 --warmup 10 --check-rows 20000 > gpurun_out/var.json 2> gpurun_out/var.err || { echo "$lib FAILED"; tail -n 5 gpurun_out/var.err; continue; }
  python - "$lib" <<'PY'
import json,sys
d=json.load(open('gpurun_out/var.json'))
print(sys.argv[1].split('/')[-1], 'ms_per_step %.4f kernel %.4f around_us %.1f exact %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], (d['ms_per_step']-d['roofline']['kernel_ms'])*1e3, d['parity']['ids_bit_exact'] and d['parity']['scores_bit_exact']))
PY
done
done
timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py tests/test_search_gpu.py -q -m gpu -x --timeout 600 -p no:cacheprovider 2>&1 | tail -n 5
timeout -k 10 200 python tools/gemm_bench.py 65536 102 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python bench.py --legs embed --no-cpu-baseline --steps 5 --warmup 2 2> gpurun_out/var2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['embed']; print('embed', e['value'], e['roofline']['frac'], json.dumps(e.get('ceiling'))[:1500])"
