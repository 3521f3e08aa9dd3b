set -e
for cfg in "--lanes 3" "--lanes 5" "--lanes 8" "--lanes 5 --seed-tiles 2048" "--lanes 8 --seed-tiles 2048" "--no-overlap --seed-tiles 2048"; do
  echo "== $cfg" >> gpurun_out/ovl_sweep.txt
  timeout -k 10 200 python bench.py --steps 50 --warmup 10 --legs none --check-rows 0 --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.0f ms_per_step %.4f kernel %.4f serial %s stepdev %s cands %s'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],(d.get('serial') or {}).get('ms_per_step'),d['step_ms_device'],d['search_stats'].get('candidates')))" >> gpurun_out/ovl_sweep.txt
done
