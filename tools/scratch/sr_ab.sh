set -o pipefail
for rnd in 1 2 3; do
  for b in 0 1; do
    CODERAG_HIP_I8_SAMPLE_RECORD=$b timeout -k 10 120 python3 bench.py --legs none --steps 200 --warmup 20 --no-cpu-baseline --check-rows 1000000 2> gpurun_out/sr_$b.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sample_record=$b round $rnd: ms/step %.4f  kernel_ms %.4f  frac %.4f  step_frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['whole_step_frac']), d['step_ms_device'], 'parity', d['parity']['ids_bit_exact'], d['parity']['scores_bit_exact'], 'fallback', d['fallback_used'])" || { tail -5 gpurun_out/sr_$b.err; exit 1; }
  done
done
