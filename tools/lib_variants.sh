#!/bin/bash
# variants of k_scan_i8 (ring depth), one box, headline bench each, interleaved
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in $VARIANTS; do
CODERAG_HIP_LIB=$GRAFT_REPO_ROOT/code-rag_amd/lib/libcoderag_hip_$v.so timeout -k 10 300 python bench.py --legs none --steps 50 --warmup 10 > gpurun_out/r3v_$v.json 2> gpurun_out/r3v_$v.err || { tail -n 5 gpurun_out/r3v_$v.err; exit 1; }
python - $v <<'PY'
import json,sys
d=json.load(open('gpurun_out/r3v_%s.json' % sys.argv[1]))
print(sys.argv[1],'value %.0f ms_per_step %.4f kernel_ms %.4f diff_us %.1f' % (d['value'], d['ms_per_step'],d['roofline']['kernel_ms'],(d['ms_per_step']-d['roofline']['kernel_ms'])*1e3), d['parity']['ids_bit_exact'], d['parity']['scores_bit_exact'])
PY
done
